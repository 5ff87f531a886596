#include "rccl_transport.h"

#ifdef RPDE_EMU
namespace rpde {
void rccl_unique_id(char*) { fail("the emulation build has no RCCL transport"); }
RcclComm* rccl_comm_create(int, int, const char*) { fail("the emulation build has no RCCL transport"); return nullptr; }
void rccl_comm_destroy(RcclComm*) {}
void rccl_alltoallv(RcclComm*, const double*, const int64_t*, double*, const int64_t*, Stream&) {
  fail("the emulation build has no RCCL transport");
}
}  // namespace rpde
#else
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>

namespace rpde {
namespace {
struct Api {
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

const Api& api() {
  static Api a;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = nullptr;
    const char* env = std::getenv("RPDE_RCCL_LIB");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (!n) continue;
      // RTLD_GLOBAL | NOLOAD first: share the copy a host process (e.g. torch) may already hold
      h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
      if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    RPDE_REQUIRE(h != nullptr, "cannot load librccl (set RPDE_RCCL_LIB)");
    auto sym = [&](const char* s) {
      void* p = dlsym(h, s);
      RPDE_REQUIRE(p != nullptr, std::string("librccl lacks ") + s);
      return p;
    };
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.Send = reinterpret_cast<decltype(a.Send)>(sym("ncclSend"));
    a.Recv = reinterpret_cast<decltype(a.Recv)>(sym("ncclRecv"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
  });
  return a;
}

void check(ncclResult_t r, const char* what) {
  if (r != ncclSuccess) fail(std::string("RCCL: ") + what + ": " + api().GetErrorString(r));
}
}  // namespace

struct RcclComm {
  ncclComm_t comm = nullptr;
  int rank = 0, size = 1;
};

static_assert(sizeof(ncclUniqueId) == kRcclIdBytes, "ncclUniqueId size");

void rccl_unique_id(char out[kRcclIdBytes]) {
  ncclUniqueId id;
  check(api().GetUniqueId(&id), "ncclGetUniqueId");
  std::memcpy(out, &id, kRcclIdBytes);
}

RcclComm* rccl_comm_create(int rank, int size, const char id_bytes[kRcclIdBytes]) {
  RPDE_REQUIRE(size >= 1 && rank >= 0 && rank < size, "bad rank / size");
  ncclUniqueId id;
  std::memcpy(&id, id_bytes, kRcclIdBytes);
  auto* c = new RcclComm;
  c->rank = rank; c->size = size;
  const ncclResult_t r = api().CommInitRank(&c->comm, size, id, rank);
  if (r != ncclSuccess) { delete c; check(r, "ncclCommInitRank"); }
  return c;
}

void rccl_comm_destroy(RcclComm* c) {
  if (!c) return;
  if (c->comm) (void)api().CommDestroy(c->comm);
  delete c;
}

void rccl_alltoallv(RcclComm* c, const double* send, const int64_t* sc, double* recv, const int64_t* rc,
                    Stream& st) {
  const Api& a = api();
  // the block a rank keeps (the diagonal block of the pencil transpose: 1 / P of the array) is a device-to-device copy on the
  // same stream, not a send to itself through the communicator; the P - 1 others go out as one group, one message per peer
  size_t so = 0, ro = 0, self_so = 0, self_ro = 0;
  for (int q = 0; q < c->rank; ++q) { self_so += (size_t)sc[q]; self_ro += (size_t)rc[q]; }
  RPDE_REQUIRE(sc[c->rank] == rc[c->rank], "alltoallv: a rank sends itself what it receives from itself");
  if (sc[c->rank] > 0)
    RPDE_HIP(hipMemcpyAsync(recv + self_ro, send + self_so, (size_t)sc[c->rank] * sizeof(double), hipMemcpyDeviceToDevice, st.s));
  if (c->size == 1) return;
  check(a.GroupStart(), "ncclGroupStart");
  for (int q = 0; q < c->size; ++q) {
    if (q != c->rank) {
      if (sc[q] > 0) check(a.Send(send + so, (size_t)sc[q], ncclDouble, q, c->comm, st.s), "ncclSend");
      if (rc[q] > 0) check(a.Recv(recv + ro, (size_t)rc[q], ncclDouble, q, c->comm, st.s), "ncclRecv");
    }
    so += (size_t)sc[q]; ro += (size_t)rc[q];
  }
  check(a.GroupEnd(), "ncclGroupEnd");
}

}  // namespace rpde
#endif
