// gple_host.h — common part of the header-only C++ adapters that re-create the reference's kernel classes on top of the
// C-ABI (include/gple.h).  See INTEGRATION.md.
//
// The adapters are written against the reference's OWN environment: kernel.h / complex_kernel.h / predict.h here start with
// `#include "stdafx.h"` (and predict.h with "storage.h") exactly like the files they replace, and use the global names those
// headers define (NumPES, PhaseDim, PhasePoints, ClassicalPhaseVector, Eigen::VectorXd, QuantumStorage, ...).  Nothing of
// stdafx.h:107-155 or storage.h is redefined here, and nothing is pulled into the global namespace: only the reference's own
// class / function / alias names of kernel.h, complex_kernel.h and predict.h are declared at global scope, everything else
// lives in namespace gple_host.  Only a conservative subset of the Eigen interface is used (constructors by size, data(),
// rows(), cols(), size(), operator[] / operator()), so the same code compiles against Eigen and against the test scaffolding
// of tests/cpp/ref_env/ (this image has no Eigen).
#ifndef GPLE_HOST_H
#define GPLE_HOST_H

#include <cstdlib>
#include <stdexcept>
#include <string>

#include "../../include/gple.h"

namespace gple_host
{
	inline void check(int status, gple_ctx* ctx)
	{
		if (status != GPLE_OK)
			throw std::runtime_error(std::string("gple: ") + gple_status_string(status) + (ctx ? std::string(": ") + gple_ctx_last_error(ctx) : ""));
	}
	// One context per process (device from GPLE_DEVICE, default 0).  The reference's objects have no notion of a device;
	// constructing a kernel object is the hot path.  Kernel objects hold their fit handle, and a fit handle
	// keeps the context alive (include/gple.h, lifetime rule), so objects with static storage duration may outlive this holder.
	inline gple_ctx* context()
	{
		struct Holder
		{
			gple_ctx* ctx = nullptr;
			Holder()
			{
				const char* d = std::getenv("GPLE_DEVICE");
				check(gple_ctx_create(d ? std::atoi(d) : 0, nullptr, &ctx), nullptr);
			}
			~Holder() { gple_ctx_destroy(ctx); }
		};
		static Holder h;
		return h.ctx;
	}
	inline unsigned flags(bool err, bool avg, bool der)
	{
		return (err ? GPLE_CALC_ERROR : 0u) | (avg ? GPLE_CALC_AVERAGE : 0u) | (der ? GPLE_CALC_DERIVATIVE : 0u);
	}
} // namespace gple_host
#endif // GPLE_HOST_H
