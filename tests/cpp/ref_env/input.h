// input.h — TEST SCAFFOLDING standing where the reference's input.h stands: same include guard, the getters of
// InitialParameters (gaussian_process_liouville_equation/input.h:80-118) that the Optimization adapter reads, written from that
// interface.  The reference derives rmin / rmax / sigma_r0 from its input file (input.cpp); here the test sets them directly.
#ifndef INPUT_H
#define INPUT_H

#include "stdafx.h"

class InitialParameters final
{
public:
	InitialParameters(const ClassicalVector<double>& Mass, const ClassicalPhaseVector& R0, const ClassicalPhaseVector& SigmaR0_, const ClassicalPhaseVector& RMin,
		const ClassicalPhaseVector& RMax):
		mass(Mass), r0(R0), rmin(RMin), rmax(RMax), SigmaR0(SigmaR0_)
	{
	}
	const ClassicalVector<double>& get_mass(void) const { return mass; }
	const ClassicalPhaseVector& get_r0(void) const { return r0; }
	const ClassicalPhaseVector& get_rmin(void) const { return rmin; }
	const ClassicalPhaseVector& get_rmax(void) const { return rmax; }
	const ClassicalPhaseVector& get_sigma_r0(void) const { return SigmaR0; }

private:
	ClassicalVector<double> mass;
	ClassicalPhaseVector r0, rmin, rmax, SigmaR0;
};

#endif // !INPUT_H
