// storage.h — TEST SCAFFOLDING standing where the reference's storage.h stands: same include guard, same public interface
// (gaussian_process_liouville_equation/storage.h: calculate_offdiagonal_index, QuantumStorage, PhaseSpacePoint with structured
// binding support, ElementPoints, AllPoints), written from that interface so that the adapters' TrainingKernels meets the base
// class it will meet inside the reference.
#ifndef STORAGE_H
#define STORAGE_H

#include "stdafx.h"

inline constexpr std::size_t calculate_offdiagonal_index(const std::size_t RowIndex, const std::size_t ColIndex)
{
	assert(RowIndex < NumPES && ColIndex < RowIndex);
	return RowIndex * (RowIndex - 1) / 2 + ColIndex;
}

template <typename T>
using ValOrCRef = std::conditional_t<sizeof(T) <= 2 * sizeof(void*), T, const T&>;

template <typename DiagDT, typename OffDiagDT = DiagDT>
class QuantumStorage
{
public:
	using DiagonalArrayType = std::array<DiagDT, NumPES>;
	using OffDiagonalArrayType = std::array<OffDiagDT, NumOffDiagonalElements>;
	QuantumStorage() = default;
	QuantumStorage(DiagonalArrayType d, OffDiagonalArrayType o): dd(std::move(d)), od(std::move(o)) {}
	QuantumStorage(ValOrCRef<DiagDT> d, ValOrCRef<OffDiagDT> o)
	{
		dd.fill(d);
		od.fill(o);
	}
	DiagonalArrayType& get_diagonal_data() { return dd; }
	const DiagonalArrayType& get_diagonal_data() const { return dd; }
	OffDiagonalArrayType& get_offdiagonal_data() { return od; }
	const OffDiagonalArrayType& get_offdiagonal_data() const { return od; }
	DiagDT& operator()(const std::size_t Index) { return dd[Index]; }
	ValOrCRef<DiagDT> operator()(const std::size_t Index) const { return dd[Index]; }
	OffDiagDT& operator()(const std::size_t Row, const std::size_t Col)
	{
		if constexpr (std::is_same_v<DiagDT, OffDiagDT>)
			if (Row == Col) return dd[Row];
		return od[calculate_offdiagonal_index(Row, Col)];
	}
	ValOrCRef<OffDiagDT> operator()(const std::size_t Row, const std::size_t Col) const
	{
		if constexpr (std::is_convertible_v<DiagDT, OffDiagDT>)
			if (Row == Col) return dd[Row];
		return od[calculate_offdiagonal_index(Row, Col)];
	}

private:
	DiagonalArrayType dd;
	OffDiagonalArrayType od;
};

class PhaseSpacePoint
{
public:
	static constexpr std::size_t NumMembers = 2;
	PhaseSpacePoint() = default;
	PhaseSpacePoint(const ClassicalPhaseVector& R, const std::complex<double> DenMatElm): r(R), rho(DenMatElm) {}
	template <std::size_t I>
	decltype(auto) get()
	{
		if constexpr (I == 0) return (r);
		else return (rho);
	}
	template <std::size_t I>
	decltype(auto) get() const
	{
		if constexpr (I == 0) return (r);
		else return std::complex<double>(rho);
	}

private:
	ClassicalPhaseVector r;
	std::complex<double> rho;
};
template <>
struct std::tuple_size<PhaseSpacePoint>: std::integral_constant<std::size_t, PhaseSpacePoint::NumMembers>
{
};
template <>
struct std::tuple_element<0, PhaseSpacePoint>
{
	using type = ClassicalPhaseVector;
};
template <>
struct std::tuple_element<1, PhaseSpacePoint>
{
	using type = std::complex<double>;
};

using ElementPoints = EigenVector<PhaseSpacePoint>;
using AllPoints = QuantumStorage<ElementPoints>;

#endif // !STORAGE_H
