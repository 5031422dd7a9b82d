// Multi-GPU entry points of the C ABI (SURVEY.md §8b "dh_comm_*", §8e): RCCL over xGMI, one process per GPU.
//
// RCCL is bound at run time (dlopen of librccl.so.1 on the first dh_comm_* call): libdancehip.so keeps no link-time dependency
// on it, single-GPU users never load it, and inside a PyTorch process the copy torch already mapped is reused (same soname)
// instead of a second RCCL.  Only the types of <rccl/rccl.h> are used at compile time.
//
// What is here is what the destination-range sharded GCN layer needs (dance_amd/sharding.py does the same through
// torch.distributed; a consumer without torch uses these):
//   dh_comm_allgather_rows_f32  dense exchange: every rank's rows of S (or G) to every rank            (ncclAllGather)
//   dh_comm_allreduce_f32       dW / db: sum over ranks, in place                                     (ncclAllReduce)
//   dh_comm_halo_exchange_f32   all-to-all-v of the rows each peer asked for: grouped ncclSend / ncclRecv, P - 1 pairs
//   dh_comm_halo_spmm_f32       the layer's aggregation with the exchange hidden behind the interior rows: pack
//                               (dh_gather_rows_f32) -> exchange on the comm stream || interior rows (dh_spmm_csr_rows_f32) on
//                               the compute stream -> boundary rows once the halo has landed; events, no host sync.
#include <vector>
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include <mutex>

#include "common.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  char why[256] = "";
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.handle) break;
    }
    if (!r.handle) {
      snprintf(r.why, sizeof r.why, "librccl.so not found (%s)", dlerror());
      return;
    }
#define DH_SYM(field, sym)                                                          \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, sym));              \
  if (!r.field && !r.why[0]) snprintf(r.why, sizeof r.why, "RCCL symbol %s missing", sym)
    DH_SYM(GetUniqueId, "ncclGetUniqueId");
    DH_SYM(CommInitRank, "ncclCommInitRank");
    DH_SYM(CommDestroy, "ncclCommDestroy");
    DH_SYM(AllGather, "ncclAllGather");
    DH_SYM(AllReduce, "ncclAllReduce");
    DH_SYM(Send, "ncclSend");
    DH_SYM(Recv, "ncclRecv");
    DH_SYM(GroupStart, "ncclGroupStart");
    DH_SYM(GroupEnd, "ncclGroupEnd");
    DH_SYM(GetErrorString, "ncclGetErrorString");
#undef DH_SYM
  });
  return &r;
}

int need_rccl(const char* who) {
  Rccl* r = rccl();
  if (!r->handle || r->why[0]) return dh::fail(DH_ERR_COMM, "%s: %s", who, r->why);
  return DH_OK;
}

#define DH_NCCL(call, who)                                                                           \
  do {                                                                                               \
    ncclResult_t rc_ = (call);                                                                       \
    if (rc_ != ncclSuccess) return dh::fail(DH_ERR_COMM, "%s: RCCL: %s", who, rccl()->GetErrorString(rc_)); \
  } while (0)
#define DH_HIP(call, who)                                                                  \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) return dh::fail(DH_ERR_LAUNCH, "%s: %s", who, hipGetErrorString(e_)); \
  } while (0)

}  // namespace

struct dh_comm {
  ncclComm_t nccl = nullptr;
  int world = 1, rank = 0;
  hipEvent_t packed = nullptr, landed = nullptr;  // compute stream -> comm stream -> compute stream
};

static_assert(sizeof(ncclUniqueId) == DH_COMM_UNIQUE_ID_BYTES, "dh_comm_unique_id size");

extern "C" int dh_comm_unique_id(void* id_host) {
  if (!id_host) return dh::fail(DH_ERR_INVALID, "dh_comm_unique_id: null pointer");
  if (int rc = need_rccl("dh_comm_unique_id")) return rc;
  DH_NCCL(rccl()->GetUniqueId(static_cast<ncclUniqueId*>(id_host)), "dh_comm_unique_id");
  return DH_OK;
}

extern "C" int dh_comm_init(dh_comm_t* comm, int world, int rank, const void* unique_id_host) {
  if (!comm || !unique_id_host) return dh::fail(DH_ERR_INVALID, "dh_comm_init: null pointer");
  if (world < 1 || rank < 0 || rank >= world) return dh::fail(DH_ERR_INVALID, "dh_comm_init: rank %d of %d", rank, world);
  if (int rc = need_rccl("dh_comm_init")) return rc;
  dh_comm* c = new dh_comm;
  c->world = world;
  c->rank = rank;
  ncclUniqueId id;
  memcpy(&id, unique_id_host, sizeof id);
  ncclResult_t rc = rccl()->CommInitRank(&c->nccl, world, id, rank);
  if (rc != ncclSuccess) {
    delete c;
    return dh::fail(DH_ERR_COMM, "dh_comm_init: RCCL: %s", rccl()->GetErrorString(rc));
  }
  if (hipEventCreateWithFlags(&c->packed, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->landed, hipEventDisableTiming) != hipSuccess) {
    rccl()->CommDestroy(c->nccl);
    delete c;
    return dh::fail(DH_ERR_LAUNCH, "dh_comm_init: hipEventCreate failed");
  }
  *comm = c;
  return DH_OK;
}

extern "C" int dh_comm_destroy(dh_comm_t comm) {
  if (!comm) return DH_OK;
  if (comm->packed) (void)hipEventDestroy(comm->packed);
  if (comm->landed) (void)hipEventDestroy(comm->landed);
  if (comm->nccl) rccl()->CommDestroy(comm->nccl);
  delete comm;
  return DH_OK;
}

extern "C" int dh_comm_world(dh_comm_t comm) { return comm ? comm->world : 0; }
extern "C" int dh_comm_rank(dh_comm_t comm) { return comm ? comm->rank : -1; }

extern "C" int dh_comm_allgather_rows_f32(dh_comm_t comm, const float* local, int64_t rows_per_rank, int64_t width, float* out,
                                          dh_stream_t stream) {
  if (!comm) return dh::fail(DH_ERR_INVALID, "dh_comm_allgather_rows_f32: null communicator");
  if (rows_per_rank < 0 || width < 0) return dh::fail(DH_ERR_INVALID, "dh_comm_allgather_rows_f32: negative size");
  if (rows_per_rank == 0 || width == 0) return DH_OK;
  if (!local || !out) return dh::fail(DH_ERR_INVALID, "dh_comm_allgather_rows_f32: null buffer");
  DH_NCCL(rccl()->AllGather(local, out, (size_t)(rows_per_rank * width), ncclFloat, comm->nccl, dh::as_stream(stream)),
          "dh_comm_allgather_rows_f32");
  return DH_OK;
}

extern "C" int dh_comm_allreduce_f32(dh_comm_t comm, float* buf, int64_t count, dh_stream_t stream) {
  if (!comm) return dh::fail(DH_ERR_INVALID, "dh_comm_allreduce_f32: null communicator");
  if (count < 0) return dh::fail(DH_ERR_INVALID, "dh_comm_allreduce_f32: negative size");
  if (count == 0) return DH_OK;
  if (!buf) return dh::fail(DH_ERR_INVALID, "dh_comm_allreduce_f32: null buffer");
  DH_NCCL(rccl()->AllReduce(buf, buf, (size_t)count, ncclFloat, ncclSum, comm->nccl, dh::as_stream(stream)), "dh_comm_allreduce_f32");
  return DH_OK;
}

// The plan of one all-to-all-v: where peer p's block starts in the packed send buffer and in the receive buffer (row units, blocks
// ordered by peer rank), and the validity rules of the counts.  Host arithmetic only — exported so that the plan can be checked without
// a GPU (tests/test_comm_plan.py simulates the grouped send / receive pairs of a whole world from these offsets).
extern "C" int dh_comm_halo_offsets(int world, int rank, const int64_t* send_rows_host, const int64_t* recv_rows_host,
                                    int64_t* send_offset_host, int64_t* recv_offset_host, int64_t* n_send, int64_t* n_recv) {
  const char* me = "dh_comm_halo_offsets";
  if (world < 1 || rank < 0 || rank >= world) return dh::fail(DH_ERR_INVALID, "%s: bad world / rank", me);
  if (!send_rows_host || !recv_rows_host || !send_offset_host || !recv_offset_host) return dh::fail(DH_ERR_INVALID, "%s: null pointer", me);
  int64_t so = 0, ro = 0;
  for (int p = 0; p < world; ++p) {
    if (send_rows_host[p] < 0 || recv_rows_host[p] < 0) return dh::fail(DH_ERR_INVALID, "%s: negative count", me);
    if (p == rank && (send_rows_host[p] || recv_rows_host[p])) return dh::fail(DH_ERR_INVALID, "%s: a rank does not exchange rows with itself", me);
    send_offset_host[p] = so;
    recv_offset_host[p] = ro;
    so += send_rows_host[p];
    ro += recv_rows_host[p];
  }
  if (n_send) *n_send = so;
  if (n_recv) *n_recv = ro;
  return DH_OK;
}

extern "C" int dh_comm_halo_exchange_f32(dh_comm_t comm, const float* send, const int64_t* send_rows_host, float* recv,
                                         const int64_t* recv_rows_host, int64_t width, dh_stream_t stream) {
  if (!comm) return dh::fail(DH_ERR_INVALID, "dh_comm_halo_exchange_f32: null communicator");
  if (!send_rows_host || !recv_rows_host || width < 0) return dh::fail(DH_ERR_INVALID, "dh_comm_halo_exchange_f32: bad argument");
  hipStream_t st = dh::as_stream(stream);
  std::vector<int64_t> soff((size_t)comm->world), roff((size_t)comm->world);
  int64_t so = 0, ro = 0;
  if (int rc = dh_comm_halo_offsets(comm->world, comm->rank, send_rows_host, recv_rows_host, soff.data(), roff.data(), &so, &ro)) return rc;
  if ((so && !send) || (ro && !recv)) return dh::fail(DH_ERR_INVALID, "dh_comm_halo_exchange_f32: null buffer");
  if (width == 0 || (so == 0 && ro == 0)) return DH_OK;
  DH_NCCL(rccl()->GroupStart(), "dh_comm_halo_exchange_f32");
  for (int p = 0; p < comm->world; ++p) {  // one send / receive pair per peer, all in flight at once on the point-to-point links
    if (send_rows_host[p])
      DH_NCCL(rccl()->Send(send + soff[p] * width, (size_t)(send_rows_host[p] * width), ncclFloat, p, comm->nccl, st), "dh_comm_halo_exchange_f32");
    if (recv_rows_host[p])
      DH_NCCL(rccl()->Recv(recv + roff[p] * width, (size_t)(recv_rows_host[p] * width), ncclFloat, p, comm->nccl, st), "dh_comm_halo_exchange_f32");
  }
  DH_NCCL(rccl()->GroupEnd(), "dh_comm_halo_exchange_f32");
  return DH_OK;
}

extern "C" int dh_comm_halo_spmm_f32(dh_comm_t comm, int64_t n_local, int64_t n_halo, int64_t width, const int32_t* rowptr,
                                     const int32_t* col, const float* val, float* operand, int64_t ldz, const int32_t* send_idx,
                                     const int64_t* send_rows_host, const int64_t* recv_rows_host, float* send_buf,
                                     const int32_t* interior_rows, int64_t n_interior, const int32_t* boundary_rows, int64_t n_boundary,
                                     float* Y, int64_t ldy, const float* bias, int act, const void* send_relu_mask,
                                     dh_stream_t compute_stream, dh_stream_t comm_stream) {
  if (!comm) return dh::fail(DH_ERR_INVALID, "dh_comm_halo_spmm_f32: null communicator");
  if (n_local < 0 || n_halo < 0 || width < 0 || n_interior < 0 || n_boundary < 0)
    return dh::fail(DH_ERR_INVALID, "dh_comm_halo_spmm_f32: negative size");
  if (!send_rows_host || !recv_rows_host) return dh::fail(DH_ERR_INVALID, "dh_comm_halo_spmm_f32: null counts");
  if (ldz % 4 != 0) return dh::fail(DH_ERR_INVALID, "dh_comm_halo_spmm_f32: the operand's rows must be 16-byte aligned (ldz %% 4 == 0)");
  if (ldz < width) return dh::fail(DH_ERR_INVALID, "dh_comm_halo_spmm_f32: ldz %lld < width %lld", (long long)ldz, (long long)width);
  hipStream_t cs = dh::as_stream(compute_stream), xs = dh::as_stream(comm_stream);
  int64_t n_send = 0, n_recv = 0;
  for (int p = 0; p < comm->world; ++p) {
    n_send += send_rows_host[p];
    n_recv += recv_rows_host[p];
  }
  if (n_recv != n_halo) return dh::fail(DH_ERR_INVALID, "dh_comm_halo_spmm_f32: receive counts sum to %lld, n_halo = %lld", (long long)n_recv, (long long)n_halo);
  // 1. pack the rows the peers asked for (own rows of the operand; optionally masked: G = dY * [Y > 0] on the way out)
  if (n_send) {
    // packed with the operand's own row stride: the exchange below moves whole ldz-float rows into the operand's tail, so a padded
    // operand (ldz > width) keeps sender and receiver strides equal (send_buf: n_send * ldz floats)
    if (int rc = dh_gather_rows_f32(n_send, width, send_idx, operand, ldz, send_relu_mask, send_buf, ldz, compute_stream)) return rc;
  }
  DH_HIP(hipEventRecord(comm->packed, cs), "dh_comm_halo_spmm_f32");
  // 2. exchange on the comm stream: halo rows land behind the own rows of the operand
  DH_HIP(hipStreamWaitEvent(xs, comm->packed, 0), "dh_comm_halo_spmm_f32");
  if (int rc = dh_comm_halo_exchange_f32(comm, send_buf, send_rows_host, operand + n_local * ldz, recv_rows_host, ldz, comm_stream)) return rc;
  DH_HIP(hipEventRecord(comm->landed, xs), "dh_comm_halo_spmm_f32");
  // 3. rows whose neighbours are all local run while the exchange is in flight
  if (n_interior) {
    if (int rc = dh_spmm_csr_rows_f32(n_interior, interior_rows, n_local + n_halo, width, rowptr, col, val, nullptr, nullptr, operand, ldz, Y,
                                      ldy, bias, act, DH_REDUCE_SUM, compute_stream))
      return rc;
  }
  // 4. the rest once the halo has landed
  DH_HIP(hipStreamWaitEvent(cs, comm->landed, 0), "dh_comm_halo_spmm_f32");
  if (n_boundary) {
    if (int rc = dh_spmm_csr_rows_f32(n_boundary, boundary_rows, n_local + n_halo, width, rowptr, col, val, nullptr, nullptr, operand, ldz, Y,
                                      ldy, bias, act, DH_REDUCE_SUM, compute_stream))
      return rc;
  }
  return DH_OK;
}
