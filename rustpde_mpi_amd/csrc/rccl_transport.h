// Native RCCL transport of the pencil exchanges: a personalised all-to-all as grouped
// ncclSend / ncclRecv on the engine's own HIP stream (no host synchronisation, no callback), over
// xGMI between the GPUs of a node.  librccl is loaded with dlopen at first use, so the library has
// no link-time dependency on it (single-GPU hosts, the emulation build).
// Replaces the MPI_Alltoallv of the reference's pencil transposes (src/field_mpi.rs:456-477).
#pragma once
#include <cstddef>
#include <cstdint>

#include "platform.h"

namespace rpde {

constexpr size_t kRcclIdBytes = 128;   // sizeof(ncclUniqueId)

struct RcclComm;
// rank 0 creates the id and hands it to the other ranks out of band (any host-side broadcast)
void rccl_unique_id(char out[kRcclIdBytes]);
// collective over all ranks; the calling thread's current device becomes the communicator's device
RcclComm* rccl_comm_create(int rank, int size, const char id[kRcclIdBytes]);
void rccl_comm_destroy(RcclComm* c);
// segment q of `send` (sc[q] doubles) goes to rank q, segment s of `recv` arrives from rank s;
// enqueued on `st`, returns without waiting
void rccl_alltoallv(RcclComm* c, const double* send, const int64_t* sc, double* recv, const int64_t* rc,
                    Stream& st);

}  // namespace rpde
