// knowhere_amd/host/shard_group.cc -- C++ host of the list-sharded multi-GPU search (include/knhip_shards.h).
//
// One worker thread per GPU inside one process; per Search(): every worker uploads the batch; the coarse quantizer is
// SHARDED BY QUERIES (rank r assigns nq / W of them with knhip_coarse_search_device, one packed all-gather gives every rank
// the whole (nq, nprobe) assignment: the replicated stage that would otherwise bound the scaling), then every
// worker scans the lists it owns (knhip_search_canonical_device over the given assignment = IndexIVF::search_preassigned,
// k + 1 CANONICAL results, no tie rule), packs its partial result into 12-byte entries, ONE all-gather
// (ncclAllGather over the RCCL communicators of ncclCommInitAll, or staged device copies), knhip_merge_topk_device; the
// reference's admission rule at the k-th boundary is then applied ONCE over all shards' candidates (knhip_tie_flag_device;
// for the flagged queries every shard's first k arrivals, one more small all-gather, knhip_tie_resolve_device), so the
// group returns the single index's answer, ties included.  Refine: every rank computes the distances of the merged
// candidates whose rows it holds, one all-gather of the distance arrays, ONE selection (knhip_refine_select_device).
// Rank 0 downloads.  Replaces faiss IndexShards::search + merge_knn_results (IndexShards.cpp:247-256).
#include "knhip_shards.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Barrier {  // reusable barrier of n threads
    explicit Barrier(int n) : n_(n) {}
    void wait() {
        std::unique_lock<std::mutex> lk(mu_);
        const int gen = gen_;
        if (++count_ == n_) {
            count_ = 0;
            gen_++;
            cv_.notify_all();
        } else {
            cv_.wait(lk, [&] { return gen != gen_; });
        }
    }
    int n_, count_ = 0, gen_ = 0;
    std::mutex mu_;
    std::condition_variable cv_;
};

struct DevMem {
    void* p = nullptr;
    size_t bytes = 0;
    int dev = 0;
    hipError_t reserve(size_t b) {
        if (b <= bytes) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        hipError_t e = hipMalloc(&p, b);
        if (e == hipSuccess) bytes = b;
        return e;
    }
    ~DevMem() {
        if (p) {
            (void)hipSetDevice(dev);
            (void)hipFree(p);
        }
    }
};

struct Rank {
    int dev = 0;
    const knhip_index* idx = nullptr;
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    DevMem q, bits, part_d, part_i, packed, gathered, all_d, all_i, out_d, out_i, ref_d, ref_i, ck, cd, keys, cdis;
    DevMem res_d, res_i, flags, arr_d, arr_i, arr_k, arr_n, all_ad, all_ai, all_ak, all_an, rdist, rdist_all;
    // raw fp32 rows this rank holds for the refine stage: row r = vector id raw_id0 + r (device pointer on `dev`)
    const float* raw = nullptr;
    int64_t raw_n = 0, raw_id0 = 0;
    const knhip_rows* rows = nullptr;  // ... or a quantised store of them (refine_type fp16 / bf16 / sq8)
};

}  // namespace

struct knhip_shard_group {
    int n = 0;
    int transport = KNHIP_SHARDS_RCCL;
    std::vector<Rank> ranks;
    std::mutex call_mu;  // one Search() at a time per group
    // RCCL transport: a collective that some ranks enqueued and others did not can never complete; the group is then
    // unusable (every later call fails at once, destroy aborts the communicators instead of draining them)
    std::atomic<bool> dead{false};
};

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& m) {
    g_err = m;
    return code;
}

// pack (dist, id) -> 12-byte entries and back: tiny kernels, compiled as HIP in this TU
__global__ void pack_kernel(const float* d, const int64_t* i, int64_t n, uint32_t* out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    out[3 * t] = __float_as_uint(d[t]);
    out[3 * t + 1] = (uint32_t)(uint64_t)i[t];
    out[3 * t + 2] = (uint32_t)((uint64_t)i[t] >> 32);
}
__global__ void fill_empty_kernel(float* d, int64_t* i, int64_t n, float worst) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    d[t] = worst;
    i[t] = -1;
}
__global__ void unpack_kernel(const uint32_t* in, int64_t n, float* d, int64_t* i) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    d[t] = __uint_as_float(in[3 * t]);
    i[t] = (int64_t)((uint64_t)in[3 * t + 1] | ((uint64_t)in[3 * t + 2] << 32));
}

// a rank's arrivals of the flagged queries as ONE block of `stride` bytes (a multiple of 8):
// [ids nflag k int64][keys nflag k int64][counts nflag int64][distances nflag k float]
__global__ void pack_arrivals_kernel(const float* ad, const int64_t* ai, const int64_t* ak, const int64_t* an, int64_t nflag,
                                     int k, unsigned char* out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ne = nflag * k;
    int64_t* oi = reinterpret_cast<int64_t*>(out);
    int64_t* ok = oi + ne;
    int64_t* on = ok + ne;
    float* od = reinterpret_cast<float*>(on + nflag);
    if (t < ne) {
        oi[t] = ai[t];
        ok[t] = ak[t];
        od[t] = ad[t];
    }
    if (t < nflag) {
        on[t] = an[t];
    }
}
__global__ void unpack_arrivals_kernel(const unsigned char* in, int64_t stride, int W, int64_t nflag, int k, float* ad,
                                       int64_t* ai, int64_t* ak, int64_t* an) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ne = nflag * k;
    if (t >= (int64_t)W * ne) return;
    const int o = (int)(t / ne);
    const int64_t e = t - (int64_t)o * ne;
    const int64_t* bi = reinterpret_cast<const int64_t*>(in + (size_t)o * stride);
    const int64_t* bk = bi + ne;
    const int64_t* bn = bk + ne;
    const float* bd = reinterpret_cast<const float*>(bn + nflag);
    ai[t] = bi[e];
    ak[t] = bk[e];
    ad[t] = bd[e];
    if (e < nflag) {
        an[(int64_t)o * nflag + e] = bn[e];
    }
}

#define SG_HIP(call)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            err = std::string(#call) + ": " + hipGetErrorString(e_);                     \
            rc = KNHIP_ERR_HIP_RUNTIME;                                                  \
            return;                                                                      \
        }                                                                                \
    } while (0)

}  // namespace

extern "C" {

int knhip_shard_group_create(int32_t n_devices, const int32_t* device_ids, int32_t transport, knhip_shard_group** out) {
    if (n_devices <= 0 || !device_ids || !out || (transport != KNHIP_SHARDS_RCCL && transport != KNHIP_SHARDS_STAGED)) {
        return fail(KNHIP_ERR_INVALID_ARGS, "shard_group_create: bad arguments");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        return fail(KNHIP_ERR_HIP_RUNTIME, "no HIP device");
    }
    auto* g = new knhip_shard_group();
    g->n = n_devices;
    g->transport = transport;
    g->ranks.resize(n_devices);
    for (int r = 0; r < n_devices; r++) {
        if (device_ids[r] < 0 || device_ids[r] >= ndev) {
            delete g;
            return fail(KNHIP_ERR_INVALID_ARGS, "shard_group_create: device id out of range");
        }
        Rank& k = g->ranks[r];
        k.dev = device_ids[r];
        for (DevMem* m : {&k.q, &k.bits, &k.part_d, &k.part_i, &k.packed, &k.gathered, &k.all_d, &k.all_i, &k.out_d, &k.out_i,
                          &k.ref_d, &k.ref_i, &k.ck, &k.cd, &k.keys, &k.cdis, &k.res_d, &k.res_i, &k.flags, &k.arr_d, &k.arr_i,
                          &k.arr_k, &k.arr_n, &k.all_ad, &k.all_ai, &k.all_ak, &k.all_an, &k.rdist, &k.rdist_all}) {
            m->dev = k.dev;
        }
        (void)hipSetDevice(k.dev);
        if (hipStreamCreateWithFlags(&k.stream, hipStreamNonBlocking) != hipSuccess) {
            delete g;
            return fail(KNHIP_ERR_HIP_RUNTIME, "stream creation failed");
        }
    }
    if (transport == KNHIP_SHARDS_RCCL) {
        for (int a = 0; a < n_devices; a++) {
            for (int b = a + 1; b < n_devices; b++) {
                if (device_ids[a] == device_ids[b]) {
                    delete g;
                    return fail(KNHIP_ERR_INVALID_ARGS, "RCCL transport needs distinct devices (use KNHIP_SHARDS_STAGED)");
                }
            }
        }
        std::vector<ncclComm_t> comms(n_devices);
        std::vector<int> devs(device_ids, device_ids + n_devices);
        const ncclResult_t nr = ncclCommInitAll(comms.data(), n_devices, devs.data());
        if (nr != ncclSuccess) {
            delete g;
            return fail(KNHIP_ERR_HIP_RUNTIME, std::string("ncclCommInitAll: ") + ncclGetErrorString(nr));
        }
        for (int r = 0; r < n_devices; r++) {
            int cnt = 0;
            (void)ncclCommCount(comms[r], &cnt);
            if (cnt != n_devices) {  // the communicator really spans the group
                delete g;
                return fail(KNHIP_ERR_HIP_RUNTIME, "RCCL communicator does not span the device list");
            }
            g->ranks[r].comm = comms[r];
        }
    }
    *out = g;
    return KNHIP_OK;
}

void knhip_shard_group_destroy(knhip_shard_group* g) {
    if (!g) return;
    for (Rank& k : g->ranks) {
        (void)hipSetDevice(k.dev);
        if (k.comm) (void)(g->dead.load() ? ncclCommAbort(k.comm) : ncclCommDestroy(k.comm));
        if (k.stream) (void)hipStreamDestroy(k.stream);
    }
    delete g;
}

int32_t knhip_shard_group_size(const knhip_shard_group* g) { return g ? g->n : 0; }

int knhip_shard_group_set_index(knhip_shard_group* g, int32_t rank, const knhip_index* idx) {
    if (!g || rank < 0 || rank >= g->n || !idx) return fail(KNHIP_ERR_INVALID_ARGS, "shard_group_set_index: bad arguments");
    g->ranks[rank].idx = idx;
    return KNHIP_OK;
}

int knhip_shard_group_set_raw(knhip_shard_group* g, int32_t rank, const float* d_rows, int64_t nrows, int64_t id_base) {
    if (!g || rank < 0 || rank >= g->n || nrows < 0 || (nrows > 0 && !d_rows)) {
        return fail(KNHIP_ERR_INVALID_ARGS, "shard_group_set_raw: bad arguments");
    }
    g->ranks[rank].raw = d_rows;
    g->ranks[rank].raw_n = nrows;
    g->ranks[rank].raw_id0 = id_base;
    return KNHIP_OK;
}

int knhip_shard_group_set_raw_rows(knhip_shard_group* g, int32_t rank, const knhip_rows* rows, int64_t id_base) {
    if (!g || rank < 0 || rank >= g->n) {
        return fail(KNHIP_ERR_INVALID_ARGS, "shard_group_set_raw_rows: bad arguments");
    }
    g->ranks[rank].rows = rows;
    g->ranks[rank].raw = nullptr;
    g->ranks[rank].raw_n = rows ? knhip_rows_count(rows) : 0;
    g->ranks[rank].raw_id0 = id_base;
    return KNHIP_OK;
}

// k_base = 0: plain search.  k_base >= k: k_base candidates per rank, merged; every rank re-ranks the merged candidates
// whose raw rows it holds; second exchange + merge of the (nq, k) partials.
static int search_impl(knhip_shard_group* g, const float* queries, int64_t nq, int32_t k, int32_t k_base, int32_t nprobe,
                       const uint8_t* bitset, int64_t bitset_nbits, int64_t* out_ids, float* out_dist, float* stage_ms) {
    if (!g || !queries || nq <= 0 || k <= 0 || !out_ids || !out_dist || (k_base != 0 && k_base < k)) {
        return fail(KNHIP_ERR_INVALID_ARGS, "shard_group_search: bad arguments");
    }
    for (const Rank& r : g->ranks) {
        if (!r.idx) return fail(KNHIP_ERR_INVALID_ARGS, "shard_group_search: a rank has no index");
    }
    const bool refine = k_base != 0;
    std::lock_guard<std::mutex> call_lk(g->call_mu);
    if (g->dead.load()) {
        return fail(KNHIP_ERR_HIP_RUNTIME, "shard group is unusable after a failed collective: destroy and recreate it");
    }
    const int W = g->n;
    knhip_desc desc{};
    if (int drc = knhip_index_get_desc(g->ranks[0].idx, &desc)) return fail(drc, knhip_last_error());
    const int32_t dim = desc.dim;
    const int32_t metric = desc.metric;
    const int32_t k1 = refine ? k_base : k;       // results of the first stage
    // the tie rule needs the (k1 + 1)-th canonical result; not covered (canonical answer, as on one index): k1 = 1024,
    // brute force with k1 >= 100 (the reference's reservoir)
    // (one place decides -- knhip_ties_rule_applies: KNHIP_TIES=canonical is honoured here as on one index)
    const bool ties1 = knhip_ties_rule_applies(desc.kind, k1) != 0;
    const int32_t kk1 = ties1 ? k1 + 1 : k1;      // width of the first exchange
    const int64_t ne1 = nq * (int64_t)kk1;        // entries per rank, first exchange
    const int64_t ne = nq * (int64_t)k;           // entries of the result
    // coarse quantizer sharded by queries (IVF kinds, several ranks): rank r assigns rows [r per, (r + 1) per)
    const char* cmode = getenv("KNHIP_SHARDS_COARSE");  // "replicate": every rank assigns every query (no extra collective)
    const bool ivf = desc.kind != KNHIP_BRUTE_FORCE;
    const bool shard_coarse = W > 1 && ivf && !(cmode && cmode[0] == 'r');
    const int32_t np = !ivf ? 1 : (int32_t)std::min<int64_t>(nprobe, desc.nlist);
    const int64_t per = (nq + W - 1) / W;
    const int64_t nec = shard_coarse ? per * (int64_t)np : 0;  // entries per rank of the coarse exchange
    // bytes a rank sends in one exchange, at most: packed partials / coarse slice, a block of arrivals, refine distances
    const size_t arr_stride_max = (((size_t)nq * ((size_t)k1 * 20 + 8)) + 7) & ~(size_t)7;
    const size_t send_max = std::max<size_t>({(size_t)std::max(ne1, nec) * 12, ties1 ? arr_stride_max : 0,
                                              refine ? (size_t)nq * k1 * 4 : 0});
    // brute force: the arrival keys are row numbers; a rank's rows follow those of the ranks in front of it
    std::vector<int64_t> row_base(W, 0);
    if (!ivf) {
        for (int r = 1; r < W; r++) row_base[r] = row_base[r - 1] + knhip_index_count(g->ranks[r - 1].idx);
    }
    Barrier bar(W);
    std::vector<std::atomic<int>> rcs(W);  // (read by the peers between barriers)
    for (auto& a : rcs) a.store(KNHIP_OK);
    std::atomic<int> shared_nflag{-1};     // flagged queries of the batch: the same number on every rank that got that far
    std::vector<std::string> errs(W);
    std::vector<std::thread> th;
    Rank* R = g->ranks.data();
    const int NS = refine ? 7 : 4;  // stage_ms columns: {search, gather, merge, [refine, gather, merge,] total}
    auto worker = [&](int r) {
        int rc = KNHIP_OK;  // published to rcs[r] by post() before every barrier the peers look behind
        std::string& err = errs[r];
        auto post = [&]() {
            if (rc != KNHIP_OK) rcs[r].store(rc);
        };
        Rank& me = R[r];
        bool alive = true;  // (a failed rank still walks through every barrier)
        const uint8_t* me_bits = nullptr;
        hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        auto all_ok = [&]() {
            bool ok = true;
            for (int o = 0; o < W; o++) ok = ok && rcs[o].load() == KNHIP_OK;
            return ok;
        };
        // ---- the collective: the first `bytes` of every rank's send buffer (me.packed) -> W blocks in rank order at `dst`
        // (on the rank's device).  The send buffer is fixed: the staged transport PULLS from the peers' `packed`, so a
        // caller-chosen source would be honoured by one transport only.  Same barrier walk for every rank,
        // alive or not.
        auto allgather_bytes = [&](void* dst, size_t bytes) {
            const void* src = me.packed.p;
            if (g->transport == KNHIP_SHARDS_RCCL) {
                // agreement BEFORE the collective: every rank posts its status, all read the same verdict, and only
                // then does anyone enqueue -- a rank that failed earlier (allocation, search, merge) makes every rank skip
                // the all-gather instead of leaving the others' streams behind a collective that never completes
                post();
                bar.wait();
                const bool go = all_ok();
                bar.wait();  // (nobody posts a new status before everyone has read the verdict)
                if (!go) {
                    alive = false;
                } else {
                    const ncclResult_t nr = ncclAllGather(src, dst, bytes, ncclChar, me.comm, me.stream);
                    if (nr != ncclSuccess) {
                        // the peers may have enqueued theirs: nothing can be salvaged on these communicators
                        err = std::string("ncclAllGather: ") + ncclGetErrorString(nr);
                        rc = KNHIP_ERR_HIP_RUNTIME;
                        alive = false;
                        g->dead.store(true);
                    }
                }
                post();
                bar.wait();
            } else {
                if (alive && hipStreamSynchronize(me.stream) != hipSuccess) {
                    rc = KNHIP_ERR_HIP_RUNTIME;
                    err = "stream synchronize failed";
                    alive = false;
                }
                post();
                bar.wait();  // every rank's block is complete
                if (alive && all_ok()) {
                    for (int o = 0; o < W && alive; o++) {  // pull every rank's block (peer copies; same device: plain copies)
                        const hipError_t e = hipMemcpyPeerAsync(static_cast<char*>(dst) + (size_t)o * bytes, me.dev,
                                                                R[o].packed.p, R[o].dev, bytes, me.stream);
                        if (e != hipSuccess) {
                            rc = KNHIP_ERR_HIP_RUNTIME;
                            err = std::string("hipMemcpyPeerAsync: ") + hipGetErrorString(e);
                            alive = false;
                        }
                    }
                    if (alive && hipStreamSynchronize(me.stream) != hipSuccess) {
                        rc = KNHIP_ERR_HIP_RUNTIME;
                        alive = false;
                    }
                }
                post();
                bar.wait();  // nobody's send buffer is overwritten before everyone has read it
            }
            if (!(alive && all_ok())) alive = false;
        };
        // ---- one exchange step: pack (pd, pi) [rows][kk], all-gather of the packed partials, unpack + merge -> (od, oi);
        // e_g / e_m: events recorded before the gather and after the merge.  merge = false: the gathered (float, int64)
        // entries are only unpacked into (all_d, all_i) -- the coarse assignment; rows = entries per rank / kk
        auto exchange = [&](int32_t kk, const float* pd, const int64_t* pi, float* od, int64_t* oi, hipEvent_t e_g,
                            hipEvent_t e_m, bool merge = true, int64_t rows = -1) {
            const int64_t n = (rows < 0 ? nq : rows) * (int64_t)kk;
            auto head = [&]() {
                hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, me.stream, pd, pi, n,
                                   static_cast<uint32_t*>(me.packed.p));
                SG_HIP(hipEventRecord(e_g, me.stream));
            };
            if (alive) {
                head();
                alive = rc == KNHIP_OK;
            }
            allgather_bytes(me.gathered.p, (size_t)n * 12);
            auto tail = [&]() {
                hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)((n * W + 255) / 256)), dim3(256), 0, me.stream,
                                   static_cast<const uint32_t*>(me.gathered.p), n * W, static_cast<float*>(me.all_d.p),
                                   static_cast<int64_t*>(me.all_i.p));
                if (merge) {
                    const int mrc = knhip_merge_topk_device(metric, nq, kk, W, static_cast<const float*>(me.all_d.p),
                                                            static_cast<const int64_t*>(me.all_i.p), od, oi, me.stream);
                    if (mrc != KNHIP_OK) {
                        err = std::string("knhip_merge_topk_device: ") + knhip_last_error();
                        rc = mrc;
                        return;
                    }
                }
                SG_HIP(hipEventRecord(e_m, me.stream));
            };
            if (alive) {
                tail();
                alive = rc == KNHIP_OK;
            }
        };
        auto body = [&]() {
            SG_HIP(hipSetDevice(me.dev));
            for (auto& e : ev) SG_HIP(hipEventCreate(&e));
            SG_HIP(me.q.reserve((size_t)nq * dim * sizeof(float)));
            SG_HIP(me.part_d.reserve((size_t)ne1 * sizeof(float)));
            SG_HIP(me.part_i.reserve((size_t)ne1 * sizeof(int64_t)));
            SG_HIP(me.packed.reserve(send_max));
            SG_HIP(me.gathered.reserve(send_max * W));
            SG_HIP(me.all_d.reserve((size_t)std::max(ne1, nec) * W * sizeof(float)));
            SG_HIP(me.all_i.reserve((size_t)std::max(ne1, nec) * W * sizeof(int64_t)));
            if (ivf) {
                SG_HIP(me.keys.reserve((size_t)std::max<int64_t>(nec * W, nq * (int64_t)np) * sizeof(int64_t)));
                SG_HIP(me.cdis.reserve((size_t)std::max<int64_t>(nec * W, nq * (int64_t)np) * sizeof(float)));
            }
            if (shard_coarse) {
                SG_HIP(me.ck.reserve((size_t)nec * sizeof(int64_t)));
                SG_HIP(me.cd.reserve((size_t)nec * sizeof(float)));
            }
            SG_HIP(me.out_d.reserve((size_t)ne1 * sizeof(float)));
            SG_HIP(me.out_i.reserve((size_t)ne1 * sizeof(int64_t)));
            SG_HIP(me.res_d.reserve((size_t)nq * k1 * sizeof(float)));
            SG_HIP(me.res_i.reserve((size_t)nq * k1 * sizeof(int64_t)));
            if (ties1) SG_HIP(me.flags.reserve(((size_t)2 * nq + 1) * sizeof(int32_t)));
            if (refine) {
                SG_HIP(me.ref_d.reserve((size_t)ne * sizeof(float)));
                SG_HIP(me.ref_i.reserve((size_t)ne * sizeof(int64_t)));
                SG_HIP(me.rdist.reserve((size_t)nq * k1 * sizeof(float)));
                SG_HIP(me.rdist_all.reserve((size_t)nq * k1 * W * sizeof(float)));
            }
            SG_HIP(hipMemcpyAsync(me.q.p, queries, (size_t)nq * dim * sizeof(float), hipMemcpyHostToDevice, me.stream));
            const uint8_t* d_bits = nullptr;
            if (bitset && bitset_nbits > 0) {
                const size_t bb = (size_t)((bitset_nbits + 7) / 8);
                SG_HIP(me.bits.reserve(bb));
                SG_HIP(hipMemcpyAsync(me.bits.p, bitset, bb, hipMemcpyHostToDevice, me.stream));
                d_bits = static_cast<const uint8_t*>(me.bits.p);
            }
            me_bits = d_bits;
            SG_HIP(hipEventRecord(ev[0], me.stream));
            if (shard_coarse) {
                // my slice of the queries -> (keys, coarse distances); rows past nq stay "no list"
                const int64_t lo = std::min<int64_t>(nq, r * per), hi = std::min<int64_t>(nq, (r + 1) * per);
                SG_HIP(hipMemsetAsync(me.ck.p, 0xff, (size_t)nec * sizeof(int64_t), me.stream));  // -1
                SG_HIP(hipMemsetAsync(me.cd.p, 0, (size_t)nec * sizeof(float), me.stream));
                if (hi > lo) {
                    const int crc = knhip_coarse_search_device(me.idx, static_cast<const float*>(me.q.p) + lo * dim, hi - lo, np,
                                                               static_cast<int64_t*>(me.ck.p), static_cast<float*>(me.cd.p),
                                                               me.stream);
                    if (crc != KNHIP_OK) {
                        err = std::string("knhip_coarse_search_device: ") + knhip_last_error();
                        rc = crc;
                        return;
                    }
                }
            } else if (ivf) {
                // replicated coarse stage: every rank assigns every query (the assignment is needed explicitly: the tie
                // rule's arrival pass walks the same probes)
                const int crc = knhip_coarse_search_device(me.idx, static_cast<const float*>(me.q.p), nq, np,
                                                           static_cast<int64_t*>(me.keys.p), static_cast<float*>(me.cdis.p),
                                                           me.stream);
                if (crc != KNHIP_OK) {
                    err = std::string("knhip_coarse_search_device: ") + knhip_last_error();
                    rc = crc;
                    return;
                }
            }
        };
        auto search = [&]() {
            if (shard_coarse) {
                // (all_d, all_i) hold the W blocks of `per` rows in rank order = the assignment of rows [0, W per) >= nq
                SG_HIP(hipMemcpyAsync(me.cdis.p, me.all_d.p, (size_t)nec * W * sizeof(float), hipMemcpyDeviceToDevice, me.stream));
                SG_HIP(hipMemcpyAsync(me.keys.p, me.all_i.p, (size_t)nec * W * sizeof(int64_t), hipMemcpyDeviceToDevice, me.stream));
            }
            // canonical partials, no tie rule on the shard: the rule is applied once, after the merge
            const int src = knhip_search_canonical_device(me.idx, static_cast<const float*>(me.q.p), nq, kk1, ivf ? np : nprobe,
                                                          ivf ? static_cast<const int64_t*>(me.keys.p) : nullptr,
                                                          ivf ? static_cast<const float*>(me.cdis.p) : nullptr, me_bits,
                                                          me_bits ? bitset_nbits : 0, static_cast<int64_t*>(me.part_i.p),
                                                          static_cast<float*>(me.part_d.p), me.stream);
            if (src != KNHIP_OK) {
                err = std::string("knhip_search_canonical_device: ") + knhip_last_error();
                rc = src;
            }
        };
        body();
        alive = rc == KNHIP_OK;
        if (shard_coarse) {
            exchange(np, static_cast<const float*>(me.cd.p), static_cast<const int64_t*>(me.ck.p), nullptr, nullptr, ev[3], ev[6],
                     /*merge=*/false, /*rows=*/per);
        }
        if (alive) {
            search();
            alive = rc == KNHIP_OK;
        }
        exchange(kk1, static_cast<const float*>(me.part_d.p), static_cast<const int64_t*>(me.part_i.p),
                 static_cast<float*>(me.out_d.p), static_cast<int64_t*>(me.out_i.p), ev[1], ev[2]);
        float* res_d = static_cast<float*>(me.out_d.p);
        int64_t* res_i = static_cast<int64_t*>(me.out_i.p);
        if (ties1) {
            // ---- the reference's admission rule at the k1-th boundary, once, over all shards' candidates -----------------
            int32_t nflag = 0;
            if (alive) {
                const int frc = knhip_tie_flag_device(static_cast<const float*>(me.out_d.p), static_cast<const int64_t*>(me.out_i.p),
                                                      nq, k1, static_cast<float*>(me.res_d.p), static_cast<int64_t*>(me.res_i.p),
                                                      static_cast<int32_t*>(me.flags.p), &nflag, me.stream);
                if (frc != KNHIP_OK) {
                    err = std::string("knhip_tie_flag_device: ") + knhip_last_error();
                    rc = frc;
                    alive = false;
                } else {
                    shared_nflag.store(nflag);  // (the merged rows are the same on every rank: so is the count)
                }
            }
            post();
            bar.wait();
            nflag = shared_nflag.load();
            res_d = static_cast<float*>(me.res_d.p);
            res_i = static_cast<int64_t*>(me.res_i.p);
            if (nflag > 0) {  // (every rank takes this branch or none does)
                const size_t stride = (((size_t)nflag * ((size_t)k1 * 20 + 8)) + 7) & ~(size_t)7;
                auto arrivals = [&]() {
                    SG_HIP(me.arr_d.reserve((size_t)nflag * k1 * sizeof(float)));
                    SG_HIP(me.arr_i.reserve((size_t)nflag * k1 * sizeof(int64_t)));
                    SG_HIP(me.arr_k.reserve((size_t)nflag * k1 * sizeof(int64_t)));
                    SG_HIP(me.arr_n.reserve((size_t)nflag * sizeof(int64_t)));
                    SG_HIP(me.all_ad.reserve((size_t)W * nflag * k1 * sizeof(float)));
                    SG_HIP(me.all_ai.reserve((size_t)W * nflag * k1 * sizeof(int64_t)));
                    SG_HIP(me.all_ak.reserve((size_t)W * nflag * k1 * sizeof(int64_t)));
                    SG_HIP(me.all_an.reserve((size_t)W * nflag * sizeof(int64_t)));
                    const int arc = knhip_tie_arrivals_device(
                            me.idx, static_cast<const float*>(me.q.p), static_cast<const int32_t*>(me.flags.p), nflag,
                            static_cast<const float*>(me.out_d.p), k1, ivf ? np : nprobe,
                            ivf ? static_cast<const int64_t*>(me.keys.p) : nullptr, ivf ? static_cast<const float*>(me.cdis.p) : nullptr,
                            me_bits, me_bits ? bitset_nbits : 0, row_base[r], static_cast<float*>(me.arr_d.p),
                            static_cast<int64_t*>(me.arr_i.p), static_cast<int64_t*>(me.arr_k.p), static_cast<int64_t*>(me.arr_n.p),
                            me.stream);
                    if (arc != KNHIP_OK) {
                        err = std::string("knhip_tie_arrivals_device: ") + knhip_last_error();
                        rc = arc;
                        return;
                    }
                    const int64_t nt = (int64_t)nflag * k1;
                    hipLaunchKernelGGL(pack_arrivals_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, me.stream,
                                       static_cast<const float*>(me.arr_d.p), static_cast<const int64_t*>(me.arr_i.p),
                                       static_cast<const int64_t*>(me.arr_k.p), static_cast<const int64_t*>(me.arr_n.p),
                                       (int64_t)nflag, (int)k1, static_cast<unsigned char*>(me.packed.p));
                };
                if (alive) {
                    arrivals();
                    alive = rc == KNHIP_OK;
                }
                allgather_bytes(me.gathered.p, stride);
                auto resolve = [&]() {
                    const int64_t nt = (int64_t)W * nflag * k1;
                    hipLaunchKernelGGL(unpack_arrivals_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, me.stream,
                                       static_cast<const unsigned char*>(me.gathered.p), (int64_t)stride, W, (int64_t)nflag, (int)k1,
                                       static_cast<float*>(me.all_ad.p), static_cast<int64_t*>(me.all_ai.p),
                                       static_cast<int64_t*>(me.all_ak.p), static_cast<int64_t*>(me.all_an.p));
                    const int rrc = knhip_tie_resolve_device(
                            metric, W, static_cast<const int32_t*>(me.flags.p), nflag, k1, static_cast<const float*>(me.out_d.p),
                            static_cast<const int64_t*>(me.out_i.p), static_cast<const float*>(me.all_ad.p),
                            static_cast<const int64_t*>(me.all_ai.p), static_cast<const int64_t*>(me.all_ak.p),
                            static_cast<const int64_t*>(me.all_an.p), res_d, res_i, me.stream);
                    if (rrc != KNHIP_OK) {
                        err = std::string("knhip_tie_resolve_device: ") + knhip_last_error();
                        rc = rrc;
                    }
                };
                if (alive) {
                    resolve();
                    alive = rc == KNHIP_OK;
                }
            }
            if (alive) (void)hipEventRecord(ev[2], me.stream);  // (the first stage ends behind the rule)
        }
        if (refine) {
            // every rank holds the same merged candidates (res_i [nq][k1]); it computes the distances of those whose raw
            // rows live here (the others marked "not here"), the arrays are exchanged and every rank runs the single
            // index's selection -- candidate order, tie rule included -- on the combined distances
            auto dists = [&]() {
                if (me.raw_n == 0) {
                    SG_HIP(hipMemsetAsync(me.rdist.p, 0xff, (size_t)nq * k1 * sizeof(float), me.stream));  // "not here"
                    return;
                }
                const int rrc = me.rows
                        ? knhip_refine_rows_distances_device(metric, me.rows, me.raw_id0, static_cast<const float*>(me.q.p), nq,
                                                             res_i, k1, static_cast<float*>(me.rdist.p), me.stream)
                        : knhip_refine_distances_device(metric, dim, me.raw, me.raw_n, me.raw_id0,
                                                        static_cast<const float*>(me.q.p), nq, res_i, k1,
                                                        static_cast<float*>(me.rdist.p), me.stream);
                if (rrc != KNHIP_OK) {
                    err = std::string("knhip_refine_distances_device: ") + knhip_last_error();
                    rc = rrc;
                }
            };
            if (alive) {
                dists();
                alive = rc == KNHIP_OK;
                if (alive) {
                    (void)hipMemcpyAsync(me.packed.p, me.rdist.p, (size_t)nq * k1 * sizeof(float), hipMemcpyDeviceToDevice, me.stream);
                    (void)hipEventRecord(ev[4], me.stream);
                }
            }
            allgather_bytes(me.rdist_all.p, (size_t)nq * k1 * sizeof(float));
            auto select = [&]() {
                int src = knhip_refine_combine_device(W, nq * (int64_t)k1, static_cast<const float*>(me.rdist_all.p),
                                                      static_cast<float*>(me.rdist.p), me.stream);
                if (src == KNHIP_OK) {
                    src = knhip_refine_select_device(metric, nq, res_i, static_cast<const float*>(me.rdist.p), k1, k,
                                                     static_cast<float*>(me.ref_d.p), static_cast<int64_t*>(me.ref_i.p), me.stream);
                }
                if (src != KNHIP_OK) {
                    err = std::string("knhip_refine_select_device: ") + knhip_last_error();
                    rc = src;
                    return;
                }
                SG_HIP(hipEventRecord(ev[5], me.stream));
            };
            if (alive) {
                select();
                alive = rc == KNHIP_OK;
            }
            res_d = static_cast<float*>(me.ref_d.p);
            res_i = static_cast<int64_t*>(me.ref_i.p);
        }
        auto finish = [&]() {
            if (r == 0) {
                SG_HIP(hipMemcpyAsync(out_dist, res_d, (size_t)ne * sizeof(float), hipMemcpyDeviceToHost, me.stream));
                SG_HIP(hipMemcpyAsync(out_ids, res_i, (size_t)ne * sizeof(int64_t), hipMemcpyDeviceToHost, me.stream));
            }
            SG_HIP(hipStreamSynchronize(me.stream));
            if (stage_ms) {
                float a = 0, b = 0, c = 0, d2 = 0;
                (void)hipEventElapsedTime(&a, ev[0], ev[1]);
                (void)hipEventElapsedTime(&b, ev[1], ev[2]);
                float* o = stage_ms + (size_t)NS * r;
                if (!refine) {
                    // (merge and tie rule are inside [ev1, ev2] here: reported together as gather + 0)
                    o[0] = a;
                    o[1] = b;
                    o[2] = 0.f;
                    o[3] = a + b;
                } else {
                    (void)hipEventElapsedTime(&c, ev[2], ev[4]);
                    (void)hipEventElapsedTime(&d2, ev[4], ev[5]);
                    o[0] = a;
                    o[1] = b;
                    o[2] = 0.f;
                    o[3] = c;
                    o[4] = d2;
                    o[5] = 0.f;
                    o[6] = a + b + c + d2;
                }
            }
        };
        post();
        bar.wait();
        if (alive && all_ok()) finish();
        post();
        for (auto& e : ev) {
            if (e) (void)hipEventDestroy(e);
        }
    };
    for (int r = 0; r < W; r++) th.emplace_back(worker, r);
    for (auto& t : th) t.join();
    for (int r = 0; r < W; r++) {
        if (rcs[r].load() != KNHIP_OK) return fail(rcs[r].load(), "rank " + std::to_string(r) + ": " + errs[r]);
    }
    return KNHIP_OK;
}

int knhip_shard_group_search(knhip_shard_group* g, const float* queries, int64_t nq, int32_t k, int32_t nprobe,
                             const uint8_t* bitset, int64_t bitset_nbits, int64_t* out_ids, float* out_dist,
                             float* stage_ms) {
    return search_impl(g, queries, nq, k, 0, nprobe, bitset, bitset_nbits, out_ids, out_dist, stage_ms);
}

int knhip_shard_group_search_refine(knhip_shard_group* g, const float* queries, int64_t nq, int32_t k, int32_t k_base,
                                    int32_t nprobe, const uint8_t* bitset, int64_t bitset_nbits, int64_t* out_ids,
                                    float* out_dist, float* stage_ms) {
    if (k_base < k) return fail(KNHIP_ERR_INVALID_ARGS, "shard_group_search_refine: k_base < k");
    for (int r = 0; g && r < g->n; r++) {
        if (g->ranks[r].raw_n > 0 && !g->ranks[r].raw && !g->ranks[r].rows) return fail(KNHIP_ERR_INVALID_ARGS, "shard_group_search_refine: no raw rows");
    }
    return search_impl(g, queries, nq, k, k_base, nprobe, bitset, bitset_nbits, out_ids, out_dist, stage_ms);
}

const char* knhip_shard_group_last_error() { return g_err.c_str(); }

}  // extern "C"
