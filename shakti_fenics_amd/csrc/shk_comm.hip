// Domain-decomposition communication: ghost (halo) exchange and reduction of the per-workgroup partial
// sums across subdomains.  Replaces the MPI traffic DOLFINx/PETSc generate under
// /root/reference/source/solvers.py:179,197,229 (ghost scatter_forward, VecNorm all-reduce) --
// SURVEY.md section 2 "implicit collectives" and section 8(e).
//
// Transports (one per context):
//   RCCL      grouped ncclSend/ncclRecv per neighbour + ncclAllReduce, all enqueued on the context's
//             stream: no host synchronisation inside the Krylov loop.  librccl is dlopen()ed so that a
//             single-GPU process never needs it and a torch process shares torch's copy.
//   CALLBACK  host-staged: pack -> D2H -> user callback -> H2D.  Used by the gloo tests (several ranks on
//             one GPU or none of RCCL's preconditions) and by any host MPI a user wants to plug in.
// Receives land directly in the ghost segment of the vector: ghosts are numbered by owner rank, in
// the owner's send order, so no unpack kernel is needed.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

#include "shk_device.h"

namespace shk {

struct RcclApi {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;   // optional
};

static RcclApi g_rccl;

const char* rccl_load() {
    if (g_rccl.h) return nullptr;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return "cannot dlopen librccl.so.1";
#define SYM(field, name)                                              \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name)); \
    if (!g_rccl.field) return "librccl lacks " name
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(AllReduce, "ncclAllReduce");
    SYM(AllGather, "ncclAllGather");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.CommGetAsyncError = reinterpret_cast<decltype(g_rccl.CommGetAsyncError)>(dlsym(h, "ncclCommGetAsyncError"));
    g_rccl.h = h;
    return nullptr;
}

int rccl_unique_id(void* out128) {
    ncclUniqueId id;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return -1;
    std::memcpy(out128, &id, 128);
    return 0;
}

const char* rccl_init(Ctx* c, int rank, int nranks, const void* id128) {
    ncclUniqueId id;
    std::memcpy(&id, id128, 128);
    ncclComm_t comm = nullptr;
    ncclResult_t r = g_rccl.CommInitRank(&comm, nranks, id, rank);
    if (r != ncclSuccess) return g_rccl.GetErrorString(r);
    c->comm.nccl = comm;
    c->comm.kind = Comm::RCCL;
    c->comm.rank = rank;
    c->comm.nranks = nranks;
    return nullptr;
}

// Loop-back self-test of the RCCL call sequence the data path uses (grouped ncclSend / ncclRecv on the context's
// stream, the same again on a second stream ordered by events -- the overlapped exchange of halo_begin / halo_end --,
// ncclAllReduce, the async-error query): on a one-rank communicator the peer is the rank itself.  The one-GPU boxes
// this was built on cannot host two RCCL ranks, so this is the only way those calls ever executed there.
const char* rccl_selftest(Ctx* c) {
    if (c->comm.kind != Comm::RCCL || !c->comm.nccl) return "no RCCL communicator on this context";
    ncclComm_t comm = reinterpret_cast<ncclComm_t>(c->comm.nccl);
    const int n = 1000, me = c->comm.rank;
    std::vector<double> h(n), back(3 * n + 4, 0.0);
    for (int i = 0; i < n; ++i) h[i] = 0.5 + i;
    double* d = nullptr;
    if (hipMalloc((void**)&d, (3 * n + 4) * sizeof(double)) != hipSuccess) return "hipMalloc failed";
    hipStream_t side = nullptr;
    hipEvent_t ready = nullptr, arrived = nullptr;
    const char* err = nullptr;
    do {
        if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&arrived, hipEventDisableTiming) != hipSuccess) { err = "second stream / events"; break; }
        if (hipMemcpyAsync(d, h.data(), n * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess) { err = "H2D failed"; break; }
        if (hipMemsetAsync(d + n, 0, (2 * n + 4) * sizeof(double), c->stream) != hipSuccess) { err = "memset failed"; break; }
        g_rccl.GroupStart();
        ncclResult_t r1 = g_rccl.Send(d, n, ncclDouble, me, comm, c->stream);
        ncclResult_t r2 = g_rccl.Recv(d + n, n, ncclDouble, me, comm, c->stream);
        ncclResult_t r3 = g_rccl.GroupEnd();
        if (r1 != ncclSuccess || r2 != ncclSuccess || r3 != ncclSuccess) { err = "grouped ncclSend / ncclRecv failed"; break; }
        // second round on the side stream: forwards what the first round delivered
        if (hipEventRecord(ready, c->stream) != hipSuccess || hipStreamWaitEvent(side, ready, 0) != hipSuccess) { err = "event ordering failed"; break; }
        g_rccl.GroupStart();
        r1 = g_rccl.Send(d + n, n, ncclDouble, me, comm, side);
        r2 = g_rccl.Recv(d + 2 * n, n, ncclDouble, me, comm, side);
        r3 = g_rccl.GroupEnd();
        if (r1 != ncclSuccess || r2 != ncclSuccess || r3 != ncclSuccess) { err = "grouped ncclSend / ncclRecv on the second stream failed"; break; }
        if (hipEventRecord(arrived, side) != hipSuccess || hipStreamWaitEvent(c->stream, arrived, 0) != hipSuccess) { err = "event ordering failed"; break; }
        if (g_rccl.AllReduce(d + 2 * n, d + 3 * n, 4, ncclDouble, ncclSum, comm, c->stream) != ncclSuccess) { err = "ncclAllReduce failed"; break; }
        // round 3's calls: a float exchange that lands in place (the multigrid vectors' ghosts) and the in-place
        // ncclAllGather of bytes (the replicated level's right-hand side); the float view of d[0 .. n) is the payload
        {
            float* fl = reinterpret_cast<float*>(d);
            g_rccl.GroupStart();
            r1 = g_rccl.Send(fl, 64, ncclFloat, me, comm, c->stream);
            r2 = g_rccl.Recv(fl + 2 * n - 64, 64, ncclFloat, me, comm, c->stream);   // the last 64 floats of d[0 .. n): overwritten
            r3 = g_rccl.GroupEnd();
            if (r1 != ncclSuccess || r2 != ncclSuccess || r3 != ncclSuccess) { err = "float ncclSend / ncclRecv failed"; break; }
            char* blk = reinterpret_cast<char*>(d + n);   // in-place all-gather of this rank's block of 256 bytes
            if (g_rccl.AllGather(blk + 256 * (size_t)me, blk, 256, ncclChar, comm, c->stream) != ncclSuccess) { err = "ncclAllGather failed"; break; }
        }
        if (g_rccl.CommGetAsyncError) {
            ncclResult_t ae = ncclSuccess;
            if (g_rccl.CommGetAsyncError(comm, &ae) != ncclSuccess || (ae != ncclSuccess && ae != ncclInProgress)) { err = "asynchronous RCCL error"; break; }
        }
        if (hipMemcpyAsync(back.data(), d, (3 * n + 4) * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess) { err = "D2H failed"; break; }
        if (wait_stream(c) != hipSuccess) { err = "stream synchronisation failed (or hit the SHK_COMM_TIMEOUT_S deadline)"; break; }
        for (int i = 0; i < n && !err; ++i)
            if (back[n + i] != h[i]) err = "ncclRecv delivered other values than ncclSend sent";
        for (int i = 0; i < n && !err; ++i)
            if (back[2 * n + i] != h[i]) err = "the exchange on the second stream delivered other values than the first round left";
        for (int i = 0; i < 4 && !err; ++i)
            if (back[3 * n + i] != c->comm.nranks * h[i] && c->comm.nranks == 1) err = "ncclAllReduce returned a wrong sum";
        if (!err && std::memcmp(&back[n - 32], &back[0], 64 * sizeof(float)) != 0) err = "the float exchange delivered other bytes than were sent";
        if (!err && c->comm.nranks == 1)
            for (int i = 0; i < 32 && !err; ++i)
                if (back[n + i] != h[i]) err = "the in-place ncclAllGather changed this rank's block";
    } while (false);
    if (side) { (void)hipStreamSynchronize(side); (void)hipStreamDestroy(side); }
    if (ready) (void)hipEventDestroy(ready);
    if (arrived) (void)hipEventDestroy(arrived);
    (void)hipFree(d);
    return err;
}

// Time `reps` back-to-back message rounds of one kind on the context's stream (hipEvents), on whatever communicator the
// context has -- with the one-rank communicator of a one-GPU box the peer is the rank itself, so the result is the
// SOFTWARE FLOOR of a round (RCCL's kernel launch + protocol, no link): the lower bound of the scaling model's free
// parameter.  kind 0: grouped ncclSend + ncclRecv of n doubles; 1: ncclAllReduce of n doubles; 2: in-place ncclAllGather
// of n bytes per rank.  Returns microseconds per round, or < 0.
double rccl_time_round(Ctx* c, int kind, int64_t n, int reps) {
    if (c->comm.kind != Comm::RCCL || !c->comm.nccl || reps < 1 || n < 1) return -1.0;
    ncclComm_t comm = reinterpret_cast<ncclComm_t>(c->comm.nccl);
    const int me = c->comm.rank, R = c->comm.nranks;
    const size_t bytes = (size_t)std::max<int64_t>(n, 1) * sizeof(double) * 2 * (size_t)std::max(R, 1);
    double* d = nullptr;
    if (hipMalloc((void**)&d, bytes) != hipSuccess) return -1.0;
    (void)hipMemsetAsync(d, 0, bytes, c->stream);
    hipEvent_t a = nullptr, b = nullptr;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    auto once = [&]() {
        if (kind == 0) {
            const int peer = R > 1 ? (me ^ 1) < R ? (me ^ 1) : me : me;
            g_rccl.GroupStart();
            g_rccl.Send(d, (size_t)n, ncclDouble, peer, comm, c->stream);
            g_rccl.Recv(d + n, (size_t)n, ncclDouble, peer, comm, c->stream);
            g_rccl.GroupEnd();
        } else if (kind == 1) {
            g_rccl.AllReduce(d, d, (size_t)n, ncclDouble, ncclSum, comm, c->stream);
        } else {
            char* blk = reinterpret_cast<char*>(d);
            g_rccl.AllGather(blk + (size_t)n * me, blk, (size_t)n, ncclChar, comm, c->stream);
        }
    };
    for (int i = 0; i < 3; ++i) once();   // warm
    (void)hipEventRecord(a, c->stream);
    for (int i = 0; i < reps; ++i) once();
    (void)hipEventRecord(b, c->stream);
    float ms = -1.0f;
    if (wait_stream(c) == hipSuccess) (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    (void)hipFree(d);
    return ms < 0.0f ? -1.0 : 1e3 * (double)ms / reps;
}

void comm_abort(Ctx* c) {
    if (c->comm.kind == Comm::RCCL && c->comm.nccl) {
        typedef ncclResult_t (*abort_fn)(ncclComm_t);
        abort_fn f = g_rccl.h ? reinterpret_cast<abort_fn>(dlsym(g_rccl.h, "ncclCommAbort")) : nullptr;
        if (f) f(reinterpret_cast<ncclComm_t>(c->comm.nccl));   // else: leaked on purpose (ncclCommDestroy would wait)
    }
    c->comm.nccl = nullptr;
    c->comm.kind = Comm::NONE;
}

void comm_destroy(Ctx* c) {
    if (c->comm.kind == Comm::RCCL && c->comm.nccl && g_rccl.CommDestroy)
        g_rccl.CommDestroy(reinterpret_cast<ncclComm_t>(c->comm.nccl));
    c->comm.nccl = nullptr;
    c->comm.kind = Comm::NONE;
}

template <class T, class TB>
__global__ __launch_bounds__(kBlock) void k_pack(int64_t n, const int32_t* __restrict__ idx,
                                                 const T* __restrict__ v, TB* __restrict__ buf) {
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        buf[i] = (TB)v[idx[i]];
}
__global__ __launch_bounds__(kBlock) void k_unpack_f32(int64_t n, const double* __restrict__ buf,
                                                       float* __restrict__ ghost) {
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        ghost[i] = (float)buf[i];
}

// Send the packed buffer, receive the neighbours' into `recv` (device memory, neighbour-major).  T = double, or float
// over RCCL (a float vector's ghosts then land straight in its ghost segment; the host-staged callbacks carry doubles).
template <class T>
static hipError_t exchange_packed(Ctx* c, const HaloPlan& P, T* recv, hipStream_t stream) {
    Comm& m = c->comm;
    const int64_t nsend = P.send_ptr.back(), nrecv = P.recv_ptr.back();
    m.n_exchange += 1;
    m.bytes_exchange += nsend * (int64_t)sizeof(T);
    if (m.timing_only) return hipSuccess;   // measurement aid: the message is skipped, `recv` keeps what it held
    if (m.kind == Comm::RCCL) {
        ncclComm_t comm = reinterpret_cast<ncclComm_t>(m.nccl);
        const ncclDataType_t dt = sizeof(T) == sizeof(double) ? ncclDouble : ncclFloat;
        const T* sendbuf = reinterpret_cast<const T*>(m.d_sendbuf);
        g_rccl.GroupStart();
        for (size_t k = 0; k < P.nbr.size(); ++k) {
            const int64_t ns = P.send_ptr[k + 1] - P.send_ptr[k], nr = P.recv_ptr[k + 1] - P.recv_ptr[k];
            if (ns > 0) g_rccl.Send(sendbuf + P.send_ptr[k], (size_t)ns, dt, P.nbr[k], comm, stream);
            if (nr > 0) g_rccl.Recv(recv + P.recv_ptr[k], (size_t)nr, dt, P.nbr[k], comm, stream);
        }
        ncclResult_t r = g_rccl.GroupEnd();
        if (r == ncclSuccess && g_rccl.CommGetAsyncError) {   // a transport failure must not surface later as a hung solve
            ncclResult_t ae = ncclSuccess;
            if (g_rccl.CommGetAsyncError(comm, &ae) != ncclSuccess || (ae != ncclSuccess && ae != ncclInProgress)) r = ncclSystemError;
        }
        return r == ncclSuccess ? hipSuccess : hipErrorUnknown;
    }
    // CALLBACK: host-staged, doubles only
    if constexpr (sizeof(T) != sizeof(double)) return hipErrorInvalidValue;
    hipError_t e = hipSuccess;
    if (nsend > 0)
        e = hipMemcpyAsync(m.h_send, m.d_sendbuf, (size_t)nsend * sizeof(double), hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
    std::vector<int32_t> nb(P.nbr.begin(), P.nbr.end());
    if (m.cb_exchange(m.cb_user, (int32_t)nb.size(), nb.data(), m.h_send, P.send_ptr.data(), m.h_recv,
                      P.recv_ptr.data()) != 0)
        return hipErrorUnknown;
    if (nrecv > 0)
        e = hipMemcpyAsync((void*)recv, m.h_recv, (size_t)nrecv * sizeof(double), hipMemcpyHostToDevice, stream);
    return e;
}

// Fill the ghost segment of `vec` (a vector of the level that `P` belongs to) with the owners' current values.
static hipError_t exchange_f64(Ctx* c, const HaloPlan& P, double* vec, hipStream_t stream) {
    Comm& m = c->comm;
    const int64_t nsend = P.send_ptr.back();
    if (nsend > 0) {
        const int g = (int)std::min<int64_t>((nsend + kBlock - 1) / kBlock, 1024);
        hipLaunchKernelGGL((k_pack<double, double>), dim3(g), dim3(kBlock), 0, stream, nsend, P.d_send_idx, vec, m.d_sendbuf);
    }
    return exchange_packed(c, P, vec + P.n_own, stream);
}
// The same for a float vector of the multigrid preconditioner.  Over RCCL the values travel as floats and land
// straight in the ghost segment (one pack kernel, no unpack); through the host-staged callbacks, whose ABI carries
// doubles, they are widened on the way out and narrowed again on arrival.
static hipError_t exchange_f32(Ctx* c, const HaloPlan& P, float* vec, hipStream_t stream) {
    Comm& m = c->comm;
    const int64_t nsend = P.send_ptr.back(), nrecv = P.recv_ptr.back();
    const int gp = (int)std::min<int64_t>((nsend + kBlock - 1) / kBlock, 1024);
    if (m.kind == Comm::RCCL) {
        if (nsend > 0)
            hipLaunchKernelGGL((k_pack<float, float>), dim3(gp), dim3(kBlock), 0, stream, nsend, P.d_send_idx, vec,
                               reinterpret_cast<float*>(m.d_sendbuf));
        return exchange_packed(c, P, vec + P.n_own, stream);
    }
    if (nsend > 0)
        hipLaunchKernelGGL((k_pack<float, double>), dim3(gp), dim3(kBlock), 0, stream, nsend, P.d_send_idx, vec, m.d_sendbuf);
    hipError_t e = exchange_packed(c, P, m.d_recvbuf, stream);
    if (e != hipSuccess) return e;
    if (nrecv > 0) {
        const int g = (int)std::min<int64_t>((nrecv + kBlock - 1) / kBlock, 1024);
        hipLaunchKernelGGL(k_unpack_f32, dim3(g), dim3(kBlock), 0, stream, nrecv, m.d_recvbuf, vec + P.n_own);
    }
    return hipSuccess;
}

static bool no_exchange(const Ctx* c, const HaloPlan& P) {
    return c->comm.kind == Comm::NONE || c->comm.nranks <= 1 || P.nbr.empty();
}
hipError_t halo_exchange_plan(Ctx* c, const HaloPlan& P, double* vec) {
    if (no_exchange(c, P)) return hipSuccess;
    PhaseTimer t(c, SHK_PH_HALO);
    return exchange_f64(c, P, vec, c->stream);
}
hipError_t halo_exchange_plan_f32(Ctx* c, const HaloPlan& P, float* vec) {
    if (no_exchange(c, P)) return hipSuccess;
    PhaseTimer t(c, SHK_PH_HALO);
    return exchange_f32(c, P, vec, c->stream);
}

// Overlapped level-0 exchange (Ctx::overlap): the caller has recorded ev_ready behind the kernel that completed `vec`
// and has launched its interior pass; pack, transport and unpack run on comm_stream, ev_halo marks their end.
template <class T>
static hipError_t halo_begin_t(Ctx* c, T* vec) {
    const HaloPlan& P = c->comm.plans[0];
    hipError_t e;
    if ((e = hipStreamWaitEvent(c->comm_stream, c->ev_ready, 0)) != hipSuccess) return e;
    if (!no_exchange(c, P)) {
        c->comm.n_overlapped += 1;
        if constexpr (sizeof(T) == sizeof(double)) e = exchange_f64(c, P, vec, c->comm_stream);
        else e = exchange_f32(c, P, vec, c->comm_stream);
        if (e != hipSuccess) return e;
    }
    return hipEventRecord(c->ev_halo, c->comm_stream);
}
hipError_t halo_begin(Ctx* c, double* vec) { return halo_begin_t(c, vec); }
hipError_t halo_begin_f32(Ctx* c, float* vec) { return halo_begin_t(c, vec); }
hipError_t halo_end(Ctx* c) { return hipStreamWaitEvent(c->stream, c->ev_halo, 0); }

hipError_t halo_exchange(Ctx* c, double* vec) {
    if (c->comm.plans.empty()) return hipSuccess;
    return halo_exchange_plan(c, c->comm.plans[0], vec);
}

// dst = element-wise sum over subdomains of src (n doubles, device memory; src == dst allowed).
hipError_t allreduce_buffer(Ctx* c, const double* src, double* dst, size_t n) {
    Comm& m = c->comm;
    if (m.kind == Comm::NONE || m.nranks <= 1) {
        if (src != dst) return hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream);
        return hipSuccess;
    }
    PhaseTimer t(c, SHK_PH_HALO);
    m.n_allreduce += 1;
    m.bytes_allreduce += (int64_t)(n * sizeof(double));
    if (m.timing_only) {   // measurement aid: the sum stays local
        if (src != dst) return hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream);
        return hipSuccess;
    }
    if (m.kind == Comm::RCCL) {
        ncclResult_t r = g_rccl.AllReduce(src, dst, n, ncclDouble, ncclSum, reinterpret_cast<ncclComm_t>(m.nccl), c->stream);
        return r == ncclSuccess ? hipSuccess : hipErrorUnknown;
    }
    if (n > m.h_red_cap) {
        if (m.h_red) (void)hipHostFree(m.h_red);
        m.h_red = nullptr;
        hipError_t e = hipHostMalloc((void**)&m.h_red, n * sizeof(double));
        if (e != hipSuccess) return e;
        m.h_red_cap = n;
    }
    hipError_t e = hipMemcpyAsync(m.h_red, src, n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e != hipSuccess) return e;
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return e;
    if (m.cb_allreduce(m.cb_user, m.h_red, (int64_t)n) != 0) return hipErrorUnknown;
    return hipMemcpyAsync(dst, m.h_red, n * sizeof(double), hipMemcpyHostToDevice, c->stream);
}

// In-place all-gather of blocks of bytes: rank r's block [off[r], off[r+1]) of `buf` (device memory; this rank's own
// block is valid on entry) becomes valid on every rank.  RCCL: ncclAllGather when the blocks are equal, else grouped
// ncclSend / ncclRecv between all pairs.  Host-staged: through the EXCHANGE callback with every other rank as a
// neighbour (the block travels as doubles, padded to 8 bytes), so any transport written against the two callbacks of
// shk_comm_init_callbacks serves it.  Replaces the all-reduces of zero-padded vectors of rounds 1-2: 1/P of the bytes
// (and float instead of double for the replicated level's right-hand side and operator values).
hipError_t allgather_blocks(Ctx* c, void* buf, const std::vector<int64_t>& off) {
    Comm& m = c->comm;
    if (m.kind == Comm::NONE || m.nranks <= 1) return hipSuccess;
    PhaseTimer t(c, SHK_PH_HALO);
    const int R = m.nranks, me = m.rank;
    const int64_t mine = off[me + 1] - off[me];
    m.n_allgather += 1;
    m.bytes_allgather += off[R] - mine;
    if (m.timing_only) return hipSuccess;
    char* b = reinterpret_cast<char*>(buf);
    if (m.kind == Comm::RCCL) {
        ncclComm_t comm = reinterpret_cast<ncclComm_t>(m.nccl);
        bool equal = true;
        for (int r = 0; r < R; ++r) equal = equal && off[r + 1] - off[r] == mine;
        ncclResult_t res = ncclSuccess;
        if (equal) {
            res = g_rccl.AllGather(b + off[me], b, (size_t)mine, ncclChar, comm, c->stream);
        } else {
            g_rccl.GroupStart();
            for (int r = 0; r < R; ++r) {
                if (r == me) continue;
                if (mine > 0) g_rccl.Send(b + off[me], (size_t)mine, ncclChar, r, comm, c->stream);
                if (off[r + 1] > off[r]) g_rccl.Recv(b + off[r], (size_t)(off[r + 1] - off[r]), ncclChar, r, comm, c->stream);
            }
            res = g_rccl.GroupEnd();
        }
        return res == ncclSuccess ? hipSuccess : hipErrorUnknown;
    }
    // host-staged: my block, padded to doubles, goes to every other rank through the exchange callback
    auto nd = [&](int r) { return (off[r + 1] - off[r] + 7) / 8; };   // doubles of rank r's block
    std::vector<int32_t> nb;
    std::vector<int64_t> sp(1, 0), rp(1, 0);
    for (int r = 0; r < R; ++r) {
        if (r == me) continue;
        nb.push_back(r);
        sp.push_back(sp.back() + nd(me));
        rp.push_back(rp.back() + nd(r));
    }
    const size_t need = (size_t)std::max<int64_t>(sp.back() + rp.back(), 1);
    if (need > m.h_red_cap) {
        if (m.h_red) (void)hipHostFree(m.h_red);
        m.h_red = nullptr;
        hipError_t e = hipHostMalloc((void**)&m.h_red, need * sizeof(double));
        if (e != hipSuccess) return e;
        m.h_red_cap = need;
    }
    double* hs = m.h_red;
    double* hr = m.h_red + sp.back();
    hipError_t e = hipSuccess;
    if (mine > 0) {
        hs[nd(me) - 1] = 0.0;   // padding bytes of the last double
        e = hipMemcpyAsync(hs, b + off[me], (size_t)mine, hipMemcpyDeviceToHost, c->stream);
    }
    if (e != hipSuccess) return e;
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return e;
    for (size_t k = 1; k < nb.size(); ++k) std::memcpy(hs + sp[k], hs, (size_t)nd(me) * sizeof(double));
    if (m.cb_exchange(m.cb_user, (int32_t)nb.size(), nb.data(), hs, sp.data(), hr, rp.data()) != 0) return hipErrorUnknown;
    for (size_t k = 0; k < nb.size(); ++k) {
        const int r = nb[k];
        if (off[r + 1] > off[r] &&
            (e = hipMemcpyAsync(b + off[r], hr + rp[k], (size_t)(off[r + 1] - off[r]), hipMemcpyHostToDevice, c->stream)) != hipSuccess)
            return e;
    }
    // the pinned staging buffer is reused by the next call: the copies above must have left it
    return hipStreamSynchronize(c->stream);
}

// One workgroup per reduction slot: this subdomain's partial array summed in a fixed order (then, when the products
// ran in two passes, the boundary pass's array behind it).
__global__ __launch_bounds__(kBlock) void k_reduce_parts(int n, const double* __restrict__ part, int nb,
                                                         const double* __restrict__ part_b, double* __restrict__ out) {
    __shared__ double sh[4];
    const double* p = part + (size_t)blockIdx.x * kMaxParts;
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += kBlock) a += p[i];
    if (nb > 0) {
        const double* q = part_b + (size_t)blockIdx.x * kMaxParts;
        for (int i = threadIdx.x; i < nb; i += kBlock) a += q[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// Sum over subdomains of `nslots` consecutive reduction slots starting at slot `first`: each subdomain first sums its
// own partial arrays (fixed order), then ONE all-reduce of nslots doubles (16-32 B) completes them; every subdomain
// ends up with the same bits, so all take identical decisions.
hipError_t allreduce_parts(Ctx* c, int first, int nslots) {
    if (c->comm.kind == Comm::NONE || c->comm.nranks <= 1) return hipSuccess;
    const int nb = c->overlap ? std::min((c->n_bslices + 3) / 4, kMaxParts) : 0;   // grid of launch_spmv_boundary
    hipLaunchKernelGGL(k_reduce_parts, dim3(nslots), dim3(kBlock), 0, c->stream, c->grid,
                       c->d_part + (size_t)first * kMaxParts, nb, nb > 0 ? c->d_part_b + (size_t)first * kMaxParts : nullptr,
                       c->d_red + first);
    return allreduce_buffer(c, c->d_red + first, c->d_red + first, (size_t)nslots);
}

// The same for partial arrays outside the Krylov slots (`part`: nslots arrays of kMaxParts, c->grid entries used):
// fixed-order local sums into red[0 .. nslots), then one all-reduce.
hipError_t allreduce_part_arrays(Ctx* c, const double* part, double* red, int nslots) {
    hipLaunchKernelGGL(k_reduce_parts, dim3(nslots), dim3(kBlock), 0, c->stream, c->grid, part, 0, (const double*)nullptr, red);
    return allreduce_buffer(c, red, red, (size_t)nslots);
}

}  // namespace shk
