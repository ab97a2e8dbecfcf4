// fake_allgather.cpp — TEST SCAFFOLDING: an in-process stand-in for ncclAllGather between host THREADS of one process that share
// one GPU (a one-GPU box cannot run two RCCL ranks on the same device).  Plugged into the library through
// gple_set_allgather_function() by tests/test_gpu_sharded.py to drive gple_*_predict_sharded with world = 2 and 3; the real
// RCCL path is exercised by the same test with a one-rank communicator.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <mutex>
#include <vector>

namespace
{
	struct Group
	{
		int world = 0, arrived = 0, generation = 0;
		std::vector<const void*> send;
		std::mutex mu;
		std::condition_variable cv;
		void barrier()
		{
			std::unique_lock<std::mutex> lk(mu);
			const int gen = generation;
			if (++arrived == world)
			{
				arrived = 0;
				++generation;
				cv.notify_all();
			}
			else
				cv.wait(lk, [&] { return generation != gen; });
		}
	};
	struct Comm
	{
		Group* group;
		int rank;
	};
} // namespace

extern "C"
{
	void* fake_group_create(int world)
	{
		Group* g = new Group;
		g->world = world;
		g->send.assign(world, nullptr);
		return g;
	}
	void* fake_comm_create(void* group, int rank) { return new Comm{static_cast<Group*>(group), rank}; }
	void fake_comm_destroy(void* comm) { delete static_cast<Comm*>(comm); }
	void fake_group_destroy(void* group) { delete static_cast<Group*>(group); }
	// ncclAllGather's signature; datatype 8 = double
	int fake_allgather(const void* sendbuff, void* recvbuff, size_t sendcount, int datatype, void* comm, hipStream_t stream)
	{
		if (datatype != 8) return 4;
		Comm* c = static_cast<Comm*>(comm);
		Group* g = c->group;
		if (hipStreamSynchronize(stream) != hipSuccess) return 1; // this rank's slice is complete
		{
			std::lock_guard<std::mutex> lk(g->mu);
			g->send[c->rank] = sendbuff;
		}
		g->barrier();
		for (int r = 0; r < g->world; ++r)
			if (hipMemcpyAsync(static_cast<double*>(recvbuff) + r * sendcount, g->send[r], sendcount * sizeof(double), hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
		if (hipStreamSynchronize(stream) != hipSuccess) return 1;
		g->barrier(); // nobody's send buffer is reused before every rank has copied it
		return 0;
	}
}
