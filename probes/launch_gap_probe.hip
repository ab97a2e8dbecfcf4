// Cost of one dependent kernel launch on the GPU timeline: N empty kernels back to back on one stream, timed by events;
// the same through a captured hipGraph.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 1024) *p = 1; }
int main()
{
	hipStream_t s; hipStreamCreate(&s);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int n = 1000;
	for (int rep = 0; rep < 2; ++rep)
	{
		hipEventRecord(e0, s);
		for (int i = 0; i < n; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, nullptr);
		hipEventRecord(e1, s); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		printf("stream launches: %.2f us per empty kernel\n", ms * 1000 / n);
	}
	hipGraph_t g; hipGraphExec_t ge;
	hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
	for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, nullptr);
	hipStreamEndCapture(s, &g);
	hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
	for (int rep = 0; rep < 2; ++rep)
	{
		hipEventRecord(e0, s);
		for (int i = 0; i < 20; ++i) hipGraphLaunch(ge, s);
		hipEventRecord(e1, s); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		printf("graph of 50 empty kernels: %.2f us per kernel\n", ms * 1000 / (20 * 50));
	}
	return 0;
}
