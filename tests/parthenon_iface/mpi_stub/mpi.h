// single-process stand-in for the handful of MPI calls adapters/parthenon/jaybenne_amd_tasks.cpp makes
// (tests/parthenon_adapter_test.cpp plays ONE rank of N: collectives see this rank's contribution only).
#ifndef JB_IFACE_MPI_STUB_H_
#define JB_IFACE_MPI_STUB_H_
#include <cstring>
typedef int MPI_Comm;
typedef int MPI_Datatype;
typedef int MPI_Op;
#define MPI_COMM_WORLD 0
#define MPI_SUM 0
#define MPI_LONG_LONG 8
#define MPI_INT64_T 8
#define MPI_DOUBLE 8
inline int MPI_Allreduce(const void *in, void *out, int n, MPI_Datatype t, MPI_Op, MPI_Comm) { std::memcpy(out, in, (size_t)n * t); return 0; }
inline int MPI_Alltoall(const void *in, int n, MPI_Datatype t, void *out, int, MPI_Datatype, MPI_Comm) { std::memcpy(out, in, (size_t)n * t); return 0; }
inline int MPI_Alltoallv(const void *, const int *, const int *, MPI_Datatype, void *, const int *, const int *, MPI_Datatype, MPI_Comm) { return 0; }
#endif
