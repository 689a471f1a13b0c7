// knowhere_amd/host/hip_index_node.cc -- the Knowhere IndexNode of the MI355X backend.
//
// Modelled on GpuCuvsIndexNode (reference src/index/gpu_cuvs/gpu_cuvs.h:73-324) and registered the same way
// (src/index/gpu_cuvs/gpu_cuvs_ivf_pq.cc:27-63: KNOWHERE_REGISTER_GLOBAL_WITH_THREAD_POOL, IndexNodeThreadPoolWrapper
// bounding the searches in flight) under NEW index names:
//     GPU_HIP_BRUTE_FORCE, GPU_HIP_IVF_FLAT, GPU_HIP_IVF_PQ, GPU_HIP_IVF_SQ8
// Virtuals are signature-identical to include/knowhere/index/index_node.h:100-400 (shared_ptr / unique_ptr<Config>,
// milvus::OpContext*, use_knowhere_build_pool).  The node owns NO arithmetic and holds NO host copy of the base:
// everything numeric goes through the C ABI of libknhip.so (include/knhip.h) and lives in HBM.  Mapping:
//   Train   IvfIndexNode::Train (src/index/ivf/ivf.cc:547-807): MatchNlist (>= 39 points per centroid, :478-489),
//           then knhip_index_train = faiss IndexIVF::train restated on the device (k-means, PQ codebooks on residuals,
//           SQ8 ranges).  COSINE: rows normalised first (ivf.cc:559-565, NormalizeVec src/common/utils.cc:60-82).
//   Add     IvfIndexNode::Add (ivf.cc:811-844) -> knhip_index_add = IndexIVF::add_core (assign, encode the residual,
//           append); ids are the running row numbers; may be called repeatedly.  Build-time `refine` keeps the raw fp32
//           rows in a second device-resident store (IndexRefineFlat, ivf.cc:673-700).
//   Search  GpuCuvsIndexNode::Search (gpu_cuvs.h:121-190): typed config, out-id bitset materialisation (:143-162),
//           all-filtered shortcut (:163-173), knhip_search -> GenResultDataSet (takes ownership of two new[] arrays),
//           MapSearchResultIdsToOutIds.  `refine_k` = IndexRefine k_factor (ivf.cc:1080-1092): the base index is searched
//           for k * refine_k candidates, re-ranked exactly on the device (knhip_search_refine).
//   RangeSearch  IvfIndexNode::RangeSearch (ivf.cc:1231-1497) -> knhip_range_search; range_filter applied here
//           (src/common/range_util.cc:27-48).  GetIndexMeta: not_implemented, as the cuVS node (gpu_cuvs.h:192-201).
//   Devices cuvs_knowhere_index (src/common/cuvs/integration/cuvs_knowhere_index.cuh): every index instance owns a device --
//           round-robin over the visible ones at Train (select_device_id, :414-426), the one with the most free
//           memory at Deserialize (:678-690) -- or the `gpu_id` of the config.  `gpu_ids` with several ordinals deals
//           the inverted lists (FLAT: the rows) over those devices: one knhip_index per device, Search() through
//           knhip_shard_group_* (include/knhip_shards.h), bit-identical to the single-device index.
//   Serialize / Deserialize: one named blob (Type()) in a BinarySet (ivf.cc:1717-1834) in the FAISS byte format the
//           CPU nodes write (IxF2/IxFI/IxF9, IwFl, IwSq, IwPQ, IxRF; faiss_io.h), so a CPU-built index loads here and back.
#include "hip_index_node.h"

#include "../../include/knhip.h"
#include "../../include/knhip_shards.h"
#include "faiss_io.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <atomic>
#include <numeric>
#include <shared_mutex>
#include <thread>

namespace knowhere {

size_t
HipSearchPoolSize() {
    const int n = knhip_device_count();
    return (size_t)std::max(n, 1) * hip_concurrent_size_per_device;
}

namespace {

Status
ToStatus(int rc) {
    switch (rc) {
        case KNHIP_OK: return Status::success;
        case KNHIP_ERR_INVALID_ARGS: return Status::invalid_args;
        case KNHIP_ERR_NOT_TRAINED: return Status::index_not_trained;
        case KNHIP_ERR_EMPTY_INDEX: return Status::empty_index;
        case KNHIP_ERR_NOT_IMPLEMENTED: return Status::not_implemented;
        case KNHIP_ERR_OUT_OF_MEMORY: return Status::malloc_error;
        default: return Status::cuda_runtime_error;  // reused for HIP failures
    }
}

// fvec_norm_L2sqr at the scalar level of the hook table (src/simd/distances_ref.cc:57-64: float products summed in a
// DOUBLE, rounded once on return)
float
NormL2Sqr(const float* x, int64_t d) {
    double res = 0;
    for (int64_t i = 0; i < d; i++) {
        const float p = x[i] * x[i];
        res += (double)p;
    }
    return (float)res;
}

// NormalizeVec (src/common/utils.cc:60-82): rows whose norm^2 is 0 or within FloatAccuracy (1e-5) of 1 are left alone,
// the others divided by sqrt(norm^2); returns the divisor (1 when the row was left alone)
float
NormalizeRow(float* x, int64_t d) {
    const float n2 = NormL2Sqr(x, d);
    if (n2 > 0 && std::abs(1.0f - n2) > 0.00001f) {
        const float n = std::sqrt(n2);
        for (int64_t i = 0; i < d; i++) x[i] = x[i] / n;
        return n;
    }
    return 1.0f;
}
void
NormalizeRows(float* x, int64_t n, int64_t d, float* norms = nullptr) {
    for (int64_t i = 0; i < n; i++) {
        const float nr = NormalizeRow(x + i * d, d);
        if (norms) norms[i] = nr;
    }
}
// L2NormsStorage::add (cppcontrib/knowhere/IndexCosine.cpp:236-245): what IndexFlatCosine multiplies by
float
InverseL2Norm(const float* x, int64_t d) {
    const float n2 = NormL2Sqr(x, d);
    return n2 == 0.0f ? 1.0f : (1.0f / std::sqrt(n2));
}

struct KnhipHandle {
    knhip_index* p = nullptr;
    KnhipHandle() = default;
    KnhipHandle(const KnhipHandle&) = delete;
    KnhipHandle& operator=(const KnhipHandle&) = delete;
    KnhipHandle(KnhipHandle&& o) noexcept : p(o.p) { o.p = nullptr; }
    KnhipHandle& operator=(KnhipHandle&& o) noexcept {
        if (this != &o) {
            knhip_index_destroy(p);
            p = o.p;
            o.p = nullptr;
        }
        return *this;
    }
    ~KnhipHandle() { knhip_index_destroy(p); }
    void reset() {
        knhip_index_destroy(p);
        p = nullptr;
    }
};

struct RowsHandle {
    knhip_rows* p = nullptr;
    RowsHandle() = default;
    RowsHandle(const RowsHandle&) = delete;
    RowsHandle& operator=(const RowsHandle&) = delete;
    RowsHandle(RowsHandle&& o) noexcept : p(o.p) { o.p = nullptr; }
    RowsHandle& operator=(RowsHandle&& o) noexcept {
        std::swap(p, o.p);
        return *this;
    }
    ~RowsHandle() { knhip_rows_destroy(p); }
};

// one device's part of an index: everything (single device) or the inverted lists / row range this device owns
struct Shard {
    int32_t device = 0;
    KnhipHandle idx;
    KnhipHandle raw;       // refine rows (IndexRefineFlat) whose ids are [raw_base, raw_base + count(raw))
    RowsHandle rows;       // quantised refine rows (refine_type fp16 / bf16 / sq8: IndexScalarQuantizer): ids [raw_base, ..)
    int64_t raw_base = 0;
    int64_t row_base = 0;  // sharded FLAT: id of this shard's first row
};

struct GroupHandle {
    knhip_shard_group* p = nullptr;
    ~GroupHandle() { knhip_shard_group_destroy(p); }
    void reset() {
        knhip_shard_group_destroy(p);
        p = nullptr;
    }
};

// test hook: the devices the last Train / Deserialize on this thread placed its index on (node_capi.cc)
thread_local std::vector<int32_t> g_last_placement;

// select_device_id() of the cuVS integration (cuvs_knowhere_index.cuh:414-426): one counter for every index of the process
int32_t
RoundRobinDevice(int ndev) {
    static std::atomic<int> index_counter{0};
    return (int32_t)(index_counter.fetch_add(1) % std::max(ndev, 1));
}
// ... and of its deserialize (:678-690): the device with the most free memory
int32_t
MostFreeDevice(int ndev) {
    int32_t best = 0;
    int64_t best_free = -1;
    for (int i = 0; i < ndev; i++) {
        int64_t f = 0, t = 0;
        if (knhip_device_memory(i, &f, &t) == KNHIP_OK && f > best_free) {
            best_free = f;
            best = i;
        }
    }
    return best;
}

// size-balanced deal of the inverted lists (longest-processing-time-first as knowhere_amd/sharded.py::
// partition_lists): lists sorted by length (longest first, ties by list number), each dealt to the currently lightest
// shard (ties: the shard with fewer lists, then the lowest -- empty lists spread out instead of piling up on shard 0)
std::vector<int32_t>
DealLists(const std::vector<int64_t>& sizes, int W) {
    const int64_t nlist = (int64_t)sizes.size();
    std::vector<int64_t> order((size_t)nlist);
    std::iota(order.begin(), order.end(), (int64_t)0);
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return sizes[a] > sizes[b]; });
    std::vector<int64_t> load((size_t)W, 0);
    std::vector<int64_t> cnt((size_t)W, 0);
    std::vector<int32_t> owner((size_t)nlist, 0);
    for (int64_t l : order) {
        int best = 0;
        for (int r = 1; r < W; r++) {
            if (load[r] < load[best] || (load[r] == load[best] && cnt[r] < cnt[best])) best = r;
        }
        owner[(size_t)l] = best;
        load[best] += sizes[l];
        cnt[best]++;
    }
    return owner;
}

}  // namespace

template <typename DataType, int Kind>
class HipIndexNode : public IndexNode {
    // (instantiated for fp16 / bf16 / int8 too, for the static registry entries -- StaticCreateConfig and friends, as
    // KNOWHERE_REGISTER_STATIC wants them --; an OBJECT exists for fp32 only: the other vector types come through the
    // reference's conversion wrapper, see the registrations at the end of the file)

 public:
    using knowhere_config_type =
        std::conditional_t<Kind == KNHIP_BRUTE_FORCE, HipBruteForceConfig,
                           std::conditional_t<Kind == KNHIP_IVF_FLAT, HipIvfFlatConfig,
                                              std::conditional_t<Kind == KNHIP_IVF_PQ, HipIvfPqConfig, HipIvfSqConfig>>>;

    HipIndexNode(const int32_t& /*version*/, const Object& /*object*/) {
        static_assert(std::is_same_v<DataType, fp32>, "fp16 / bf16 / int8 datasets are converted by IndexNodeDataMockWrapper");
    }
    ~HipIndexNode() override = default;

    bool
    NeedBitsetExactCount() const override {
        return true;  // the all-filtered shortcut needs the exact projected count (gpu_cuvs.h:82-85)
    }

    Status
    Train(const DataSetPtr dataset, std::shared_ptr<Config> cfg, bool /*use_knowhere_build_pool*/) override {
        if (!dataset || !dataset->GetTensor() || !cfg) return Status::invalid_args;
        if (Built()) return Status::index_already_trained;
        if (knhip_abi_version() != KNHIP_ABI_VERSION) return Status::cuda_runtime_error;  // header / library mismatch
        const auto& c = static_cast<const knowhere_config_type&>(*cfg);
        const int64_t rows = dataset->GetRows();
        dim_ = dataset->GetDim();
        if (c.dim.has_value() && c.dim.value() != dim_) return Status::invalid_args;
        const std::string metric = c.metric_type.value_or(metric::L2);
        cosine_ = IsMetricType(metric, metric::COSINE);
        metric_ = IsMetricType(metric, metric::L2) ? KNHIP_L2 : KNHIP_IP;
        metric_name_ = cosine_ ? metric::COSINE : (metric_ == KNHIP_L2 ? metric::L2 : metric::IP);
        if constexpr (Kind != KNHIP_BRUTE_FORCE) {
            // MatchNlist: silently shrink nlist so that nlist * 39 <= rows (ivf.cc:478-489)
            nlist_ = c.nlist.value();
            if (nlist_ * 39 > rows) nlist_ = std::max<int64_t>(1, rows / 39);
            default_nprobe_ = c.nprobe.value_or(8);
        }
        if constexpr (Kind == KNHIP_IVF_PQ) {
            // m = 0: the backend picks, as cuVS does for pq_dim = 0 (about dim / 2): the largest m with a fast kernel (8,
            // 16, 32, 64) that divides dim and leaves sub-vectors of at least 2 dims, else the largest divisor of dim up to
            // min(128, dim / 2).  An explicit m is honoured (any divisor of dim up to 128 with sub-vectors of at most 144
            // dims, as the reference honours any divisor) or refused, never rewritten
            const bool auto_m = c.m.value_or(0) == 0;
            auto fast = [](int64_t m) { return m == 8 || m == 16 || m == 32 || m == 64; };
            auto fits = [&](int64_t m) { return m >= 1 && m <= 128 && dim_ % m == 0 && dim_ / m <= 144; };
            if (auto_m) {
                m_ = 0;
                for (int64_t m = std::min<int64_t>(64, dim_ / 2); m >= 8 && m_ == 0; m--) {
                    if (fast(m) && fits(m)) m_ = m;
                }
                for (int64_t m = std::min<int64_t>(128, std::max<int64_t>(1, dim_ / 2)); m >= 1 && m_ == 0; m--) {
                    if (fits(m)) m_ = m;
                }
            } else {
                m_ = c.m.value();
            }
            if (!fits(m_)) return Status::invalid_args;
            nbits_ = c.nbits.value_or(8);
            if (nbits_ < 1 || nbits_ > 8) return Status::invalid_args;
        }
        if constexpr (Kind == KNHIP_IVF_PQ || Kind == KNHIP_IVF_SQ8) {
            // a refine index is built iff `refine` AND `refine_type` are given (ivf_wrapper.cc:170, :214).  refine_type
            // (ivf_config.h:64-76, refine_utils.cc:20-58): fp32 / flat = IndexRefineFlat over the raw rows; fp16 / bf16 / sq8
            // / sq6 / int8 / sq4u = IndexRefine over an IndexScalarQuantizer store of the rows (knhip_rows): every type of
            // get_sq_quantizer_type (refine_utils.cc:19-26)
            has_refine_ = c.refine.value_or(false) && c.refine_type.has_value();
            refine_rows_type_ = 0;
            if (has_refine_) {
                std::string t = c.refine_type.value();
                for (auto& ch : t) ch = (char)std::tolower((unsigned char)ch);
                if (t == "fp16") {
                    refine_rows_type_ = KNHIP_ROWS_FP16;
                } else if (t == "bf16") {
                    refine_rows_type_ = KNHIP_ROWS_BF16;
                } else if (t == "sq8") {
                    refine_rows_type_ = KNHIP_ROWS_SQ8;
                } else if (t == "sq6") {
                    refine_rows_type_ = KNHIP_ROWS_SQ6;
                } else if (t == "int8") {
                    refine_rows_type_ = KNHIP_ROWS_INT8;
                } else if (t == "sq4u") {
                    refine_rows_type_ = KNHIP_ROWS_SQ4U;
                } else if (t != "fp32" && t != "flat") {
                    LOG_KNOWHERE_ERROR_ << TypeName() << ": invalid refine type " << c.refine_type.value()
                                        << " (fp32 / flat / fp16 / bf16 / sq8 / sq6 / int8 / sq4u)";
                    return Status::invalid_args;
                }
            }
        }
        std::vector<int32_t> devs;
        if (Status st = SelectDevices(c, /*deserialize=*/false, &devs); st != Status::success) return st;
        if (Status st = CreateShards(devs); st != Status::success) return st;
        if constexpr (Kind == KNHIP_BRUTE_FORCE) {
            return Status::success;  // nothing to train
        }
        const float* x = (const float*)dataset->GetTensor();
        std::vector<float> xn;
        if (cosine_) {
            xn.assign(x, x + rows * dim_);
            NormalizeRows(xn.data(), rows, dim_);
            x = xn.data();
        }
        // the reference's clustering defaults, on the first device; the trained state is replicated on the others
        int rc = knhip_index_train(sh_[0].idx.p, rows, x, nullptr);
        if (rc == KNHIP_OK && sh_.size() > 1) rc = ReplicateTrainedState();
        if (rc == KNHIP_OK && refine_rows_type_ != 0) {
            // IndexRefine::train trains the refine index on the same rows (IndexRefine.cpp:47-51): the sq8 ranges, on the
            // first device, copied to the stores of the others
            rc = CreateRowStores();
            if (rc == KNHIP_OK) {
                // (QT_4bit_uniform + L2 takes its one range from the 1 % / 99 % quantiles: refine_utils.cc:176-180)
                rc = refine_rows_type_ == KNHIP_ROWS_SQ4U
                             ? knhip_rows_train_uniform(sh_[0].rows.p, rows, x, metric_ == KNHIP_L2 ? 2 : 0,
                                                        metric_ == KNHIP_L2 ? 0.01f : 0.f)
                             : knhip_rows_train(sh_[0].rows.p, rows, x);
            }
            if (rc == KNHIP_OK) rc = ReplicateRowRanges();
        }
        if (rc) {
            DropShards();
            return ToStatus(rc);
        }
        return Status::success;
    }

    // thread safe against concurrent Search (index_node.h:141-145): the index object serialises layout changes
    Status
    Add(const DataSetPtr dataset, std::shared_ptr<Config> /*cfg*/, bool /*use_knowhere_build_pool*/) override {
        if (!dataset || !dataset->GetTensor()) return Status::invalid_args;
        if (!Built()) return Kind == KNHIP_BRUTE_FORCE ? Status::empty_index : Status::index_not_trained;
        if (dataset->GetDim() != dim_) return Status::invalid_args;
        const int64_t rows = dataset->GetRows();
        const float* x = (const float*)dataset->GetTensor();
        std::vector<float> xn;
        std::unique_lock<std::shared_mutex> lk(rw_);
        const int64_t n0 = CountLocked();
        const int W = (int)sh_.size();
        // what is assigned / encoded (cosine on PQ / SQ8: the normalised rows) and, for the stored-norm cosine kinds
        // (FLAT / IVF_FLAT keep the RAW rows and one float per row beside them, as IndexFlatCosine -- inverse norms,
        // IndexCosine.cpp:236-245 -- and IndexIVFFlatCosine -- norms, IndexIVFFlat.cpp:516-524: assigned by the
        // normalised row -- do), what is stored
        const float* x_store = x;
        const float* x_assign = x;
        const size_t scale0 = row_scale_by_id_.size();
        if (StoredNormCosine()) {
            row_scale_by_id_.resize(scale0 + (size_t)rows);
            if constexpr (Kind == KNHIP_BRUTE_FORCE) {
                for (int64_t i = 0; i < rows; i++) row_scale_by_id_[scale0 + i] = InverseL2Norm(x + i * dim_, dim_);
            } else {
                xn.assign(x, x + rows * dim_);
                NormalizeRows(xn.data(), rows, dim_, row_scale_by_id_.data() + scale0);
                x_assign = xn.data();
            }
        } else if (cosine_) {
            xn.assign(x, x + rows * dim_);
            NormalizeRows(xn.data(), rows, dim_);
            x_store = x_assign = xn.data();
        }
        int rc = KNHIP_OK;
        if constexpr (Kind == KNHIP_BRUTE_FORCE) {
            rc = AddRowRanges(&Shard::idx, /*bases=*/&Shard::row_base, n0, rows, x_store);
        } else if (W == 1) {
            rc = StoredNormCosine() ? knhip_index_add_assigned_by(sh_[0].idx.p, rows, x_store, x_assign, nullptr)
                                    : knhip_index_add(sh_[0].idx.p, rows, x_store, nullptr);
        } else {
            rc = AddSharded(n0, rows, x_store, x_assign);
        }
        // A failure from here on may have left some shard with rows the others (or the norms / the refine store / the
        // group) do not have: Count, ids and the stores would disagree for ever after.  The node drops the index and
        // reports the error -- a caller can rebuild, it cannot be handed half an Add.
        auto broken = [&](Status st) {
            LOG_KNOWHERE_ERROR_ << TypeName() << ": Add failed after rows were appended; the index is dropped";
            DropShards();
            row_scale_by_id_.clear();
            return st;
        };
        if (rc) {
            row_scale_by_id_.resize(scale0);
            // one device, or a failure before anything was appended (assignment, routing): the index is unchanged
            if (W == 1 || CountLocked() == n0) return ToStatus(rc);
            return broken(ToStatus(rc));
        }
        if (StoredNormCosine()) {
            if (Status st = PushRowScale(); st != Status::success) return broken(st);
        }
        if (NeedRawStore() && refine_rows_type_ != 0) {
            if (!sh_[0].rows.p) return broken(Status::index_not_trained);
            // id ranges as for the fp32 rows: the first batch cut into one range per device, later batches extend the last
            if (n0 == 0) {
                for (int r = 0; r < W && rc == KNHIP_OK; r++) {
                    const int64_t lo = rows * r / W, hi = rows * (r + 1) / W;
                    sh_[r].raw_base = lo;
                    if (hi > lo) rc = knhip_rows_add(sh_[r].rows.p, hi - lo, x_store + lo * dim_);
                }
            } else {
                rc = knhip_rows_add(sh_[W - 1].rows.p, rows, x_store);
            }
            if (rc) return broken(ToStatus(rc));
            if (W > 1) {
                if (Status st = AttachRawToGroup(); st != Status::success) return broken(st);
            }
        } else if (NeedRawStore()) {
            for (auto& s : sh_) {
                if (!s.raw.p) {
                    knhip_desc rd{};
                    rd.kind = KNHIP_BRUTE_FORCE;
                    rd.metric = metric_;
                    rd.dim = (int32_t)dim_;
                    rd.device = s.device;
                    if ((rc = knhip_index_create(&rd, &s.raw.p))) return broken(ToStatus(rc));
                }
            }
            if ((rc = AddRowRanges(&Shard::raw, &Shard::raw_base, n0, rows, x_store))) return broken(ToStatus(rc));
            if (W > 1) {
                if (Status st = AttachRawToGroup(); st != Status::success) return broken(st);
            }
        }
        return Status::success;
    }

    expected<DataSetPtr>
    Search(const DataSetPtr dataset, std::unique_ptr<Config> cfg, const BitsetView& bitset,
           milvus::OpContext* op_context) const override {
        if (!Built() || Count() == 0) return expected<DataSetPtr>::Err(Status::empty_index, "index not built");
        if (!dataset || !dataset->GetTensor() || !cfg)
            return expected<DataSetPtr>::Err(Status::invalid_args, "null dataset / config");
        if (dataset->GetDim() != dim_) return expected<DataSetPtr>::Err(Status::invalid_args, "dim mismatch");
        const auto& c = static_cast<const knowhere_config_type&>(*cfg);
        const int64_t nq = dataset->GetRows();
        const int64_t k = c.k.value();
        int64_t nprobe = 1;
        if constexpr (Kind != KNHIP_BRUTE_FORCE) nprobe = c.nprobe.value_or(default_nprobe_);
        checkCancellation(op_context);
        const float* q = (const float*)dataset->GetTensor();
        std::vector<float> qn;
        if (cosine_) {  // CopyAndNormalizeVecs (ivf.cc:1068-1071)
            qn.assign(q, q + nq * dim_);
            NormalizeRows(qn.data(), nq, dim_);
            q = qn.data();
        }
        // bitset in the searched (internal) id domain: materialise an installed out-id view (gpu_cuvs.h:143-162)
        std::vector<uint8_t> in_bitset;
        const uint8_t* bits = nullptr;
        int64_t nbits = 0;
        const bool has_bitset = MaterialiseBitset(bitset, &in_bitset, &bits, &nbits);
        // every row filtered: ids -1, distances +inf, like gpu_cuvs.h:163-173
        if (has_bitset && bitset.has_known_count() && bitset.count() >= bitset.size() && (int64_t)bitset.size() >= Count()) {
            auto ids = std::make_unique<int64_t[]>(nq * k);
            auto dis = std::make_unique<float[]>(nq * k);
            std::fill_n(ids.get(), nq * k, (int64_t)-1);
            std::fill_n(dis.get(), nq * k, std::numeric_limits<float>::infinity());
            return GenResultDataSet(nq, k, ids.release(), dis.release());
        }
        auto ids = std::make_unique<int64_t[]>(nq * k);
        auto dis = std::make_unique<float[]>(nq * k);
        int rc;
        const char* err_text = nullptr;
        {
            std::shared_lock<std::shared_mutex> lk(rw_);
            // use_refine = the index carries a refine index (ivf.cc:1080-1092); k_factor = refine_k
            // with k_factor 1 (the default refine_k) the reference still re-scores the k results against the refine index
            // and re-sorts them (IndexRefine::search, k_base == k): so does this node
            int64_t kbase = k;
            bool use_refine = false;
            if constexpr (Kind == KNHIP_IVF_PQ || Kind == KNHIP_IVF_SQ8) {
                if (has_refine_ && (sh_[0].raw.p || sh_[0].rows.p) && c.refine_k.has_value()) {
                    use_refine = true;
                    const int64_t want = std::max<int64_t>(k, (int64_t)(k * c.refine_k.value()));
                    kbase = std::min<int64_t>(1024, want);
                    if (kbase < want) {  // (the first stage returns at most 1024 candidates per query: said, not hidden)
                        LOG_KNOWHERE_WARNING_ << "GPU_HIP refine: k * refine_k = " << want << " clamped to " << kbase;
                    }
                }
            }
            if (sh_.size() > 1) {
                // list-sharded: every device scans the lists it owns, one all-gather of the partial top-k, device merge
                // (include/knhip_shards.h); with refine every device re-ranks the candidates whose rows it holds
                if (use_refine) {
                    rc = knhip_shard_group_search_refine(group_.p, q, nq, (int32_t)k, (int32_t)kbase, (int32_t)nprobe, bits,
                                                         nbits, ids.get(), dis.get(), nullptr);
                } else {
                    rc = knhip_shard_group_search(group_.p, q, nq, (int32_t)k, (int32_t)nprobe, bits, nbits, ids.get(),
                                                  dis.get(), nullptr);
                }
                if (rc) err_text = knhip_shard_group_last_error();
            } else if (use_refine && sh_[0].rows.p) {
                rc = knhip_search_refine_rows(sh_[0].idx.p, sh_[0].rows.p, q, nq, (int32_t)k, (int32_t)kbase, (int32_t)nprobe,
                                              bits, nbits, ids.get(), dis.get());
            } else if (use_refine) {
                rc = knhip_search_refine(sh_[0].idx.p, sh_[0].raw.p, q, nq, (int32_t)k, (int32_t)kbase, (int32_t)nprobe, bits,
                                         nbits, ids.get(), dis.get());
            } else {
                rc = knhip_search(sh_[0].idx.p, q, nq, (int32_t)k, (int32_t)nprobe, bits, nbits, ids.get(), dis.get());
            }
        }
        if (rc) return expected<DataSetPtr>::Err(ToStatus(rc), err_text ? err_text : knhip_last_error());
        auto res = GenResultDataSet(nq, k, ids.release(), dis.release());
        this->MapSearchResultIdsToOutIds(res);
        return res;
    }

    // IvfIndexNode::RangeSearch (ivf.cc:1231-1497): radius / range_filter / max_empty_result_buckets from the config,
    // every list a candidate, results filtered to [range_filter, radius) (L2) or (radius, range_filter] (IP, COSINE)
    // (range_util.h:22-25) and returned as lims + flat arrays.
    expected<DataSetPtr>
    RangeSearch(const DataSetPtr dataset, std::unique_ptr<Config> cfg, const BitsetView& bitset,
                milvus::OpContext* op_context) const override {
        if (!Built() || Count() == 0) return expected<DataSetPtr>::Err(Status::empty_index, "index not built");
        if (!dataset || !dataset->GetTensor() || !cfg)
            return expected<DataSetPtr>::Err(Status::invalid_args, "null dataset / config");
        if (dataset->GetDim() != dim_) return expected<DataSetPtr>::Err(Status::invalid_args, "dim mismatch");
        const auto& c = static_cast<const knowhere_config_type&>(*cfg);
        const float radius = c.radius.value();
        const float range_filter = c.range_filter.value();
        int64_t max_empty = 2;
        if constexpr (Kind != KNHIP_BRUTE_FORCE) max_empty = c.max_empty_result_buckets.value_or(2);
        checkCancellation(op_context);
        const int64_t nq = dataset->GetRows();
        const float* q = (const float*)dataset->GetTensor();
        std::vector<float> qn;
        if (cosine_) {
            qn.assign(q, q + nq * dim_);
            NormalizeRows(qn.data(), nq, dim_);
            q = qn.data();
        }
        std::vector<uint8_t> in_bitset;
        const uint8_t* bits = nullptr;
        int64_t nbits = 0;
        MaterialiseBitset(bitset, &in_bitset, &bits, &nbits);
        // one result per shard.  FLAT row ranges: ascending ids, so the concatenation per query is IndexFlat's row order.
        // List-sharded IVF kinds: every shard visits ALL of its lists and reports its hits per coarse rank
        // (knhip_range_search_ranked: each shard holds every centroid, so all rank the lists alike); the reference's early
        // stop counts CONSECUTIVE lists without a hit in coarse order over all lists (IndexIVF.cpp:917-933), so the counts
        // are summed over the shards per (query, rank), the ranks walked with that rule here, and each rank's hits taken
        // from the shard that owns the list -- the emission order and the stop of one device.  (What is lost is the
        // saving of the early stop: every list is scanned.)  Queries go in slices that bound the count arrays.
        const int W = (int)sh_.size();
        const bool ranked = Kind != KNHIP_BRUTE_FORCE && W > 1;
        const int64_t slice = ranked ? std::max<int64_t>(1, ((int64_t)16 << 20) / std::max<int64_t>(nlist_, 1)) : nq;
        const bool is_ip = metric_ != KNHIP_L2;
        std::vector<size_t> res_lims((size_t)nq + 1, 0);
        std::vector<int64_t> res_ids;
        std::vector<float> res_dis;
        auto keep = [&](float v) {
            return range_filter == defaultRangeFilter ||
                   (is_ip ? (radius < v && v <= range_filter) : (range_filter <= v && v < radius));
        };
        for (int64_t q0 = 0; q0 < nq; q0 += slice) {
            const int64_t n = std::min(slice, nq - q0);
            std::vector<std::vector<int64_t>> lims((size_t)W, std::vector<int64_t>((size_t)n + 1, 0));
            std::vector<int64_t*> ids((size_t)W, nullptr);
            std::vector<float*> dis((size_t)W, nullptr);
            std::vector<int32_t*> cnt((size_t)W, nullptr);
            int rc = KNHIP_OK;
            {
                std::shared_lock<std::shared_mutex> lk(rw_);
                for (int r = 0; r < W && rc == KNHIP_OK; r++) {
                    if (knhip_index_count(sh_[r].idx.p) == 0) continue;  // (an empty shard: lims stay 0, no counts)
                    rc = ranked ? knhip_range_search_ranked(sh_[r].idx.p, q + q0 * dim_, n, radius, bits, nbits,
                                                            lims[r].data(), &ids[r], &dis[r], &cnt[r])
                                : knhip_range_search(sh_[r].idx.p, q + q0 * dim_, n, radius, (int32_t)max_empty, bits, nbits,
                                                     lims[r].data(), &ids[r], &dis[r]);
                }
            }
            auto free_all = [&]() {
                for (int r = 0; r < W; r++) {
                    knhip_free(ids[r]);
                    knhip_free(dis[r]);
                    knhip_free(cnt[r]);
                }
            };
            if (rc) {
                free_all();
                return expected<DataSetPtr>::Err(ToStatus(rc), knhip_last_error());
            }
            for (int64_t i = 0; i < n; i++) {
                if (!ranked) {
                    for (int r = 0; r < W; r++) {
                        for (int64_t j = lims[r][(size_t)i]; j < lims[r][(size_t)i + 1]; j++) {
                            if (keep(dis[r][j])) {
                                res_ids.push_back(ids[r][j]);
                                res_dis.push_back(dis[r][j]);
                            }
                        }
                    }
                } else {
                    std::vector<int64_t> ptr((size_t)W);
                    for (int r = 0; r < W; r++) ptr[(size_t)r] = lims[r][(size_t)i];
                    int64_t nempty = 0;
                    for (int64_t rank = 0; rank < nlist_; rank++) {
                        int64_t hits = 0;
                        for (int r = 0; r < W; r++) {
                            const int64_t c = cnt[r] ? cnt[r][i * nlist_ + rank] : 0;
                            for (int64_t j = ptr[(size_t)r]; j < ptr[(size_t)r] + c; j++) {
                                if (keep(dis[r][j])) {
                                    res_ids.push_back(ids[r][j]);
                                    res_dis.push_back(dis[r][j]);
                                }
                            }
                            ptr[(size_t)r] += c;
                            hits += c;
                        }
                        if (max_empty > 0) {  // (the rule of range.hip::range_plan_kernel, i.e. of the reference)
                            nempty = hits == 0 ? nempty + 1 : 0;
                            if (nempty >= max_empty) break;
                        }
                    }
                }
                res_lims[(size_t)(q0 + i) + 1] = res_ids.size();
            }
            free_all();
        }
        const size_t total = res_ids.size();
        auto out_lims = std::make_unique<size_t[]>(nq + 1);
        auto out_ids = std::make_unique<int64_t[]>(std::max<size_t>(total, 1));
        auto out_dis = std::make_unique<float[]>(std::max<size_t>(total, 1));
        std::copy(res_lims.begin(), res_lims.end(), out_lims.get());
        std::copy(res_ids.begin(), res_ids.end(), out_ids.get());
        std::copy(res_dis.begin(), res_dis.end(), out_dis.get());
        auto res = GenResultDataSet(nq, out_ids.release(), out_dis.release(), out_lims.release());
        this->MapSearchResultIdsToOutIds(res);
        return res;
    }

    expected<DataSetPtr>
    GetVectorByIds(const DataSetPtr dataset, milvus::OpContext* /*op_context*/) const override {
        if (!dataset || !dataset->GetIds()) return expected<DataSetPtr>::Err(Status::invalid_args, "null ids");
        if (!HasRawData(metric_name_)) return expected<DataSetPtr>::Err(Status::not_implemented, "no raw data");
        // (IVF_FLAT: the index's own rows through its direct map -- no second copy of the raw vectors)
        if (!Built()) return expected<DataSetPtr>::Err(Status::empty_index, "index not built");
        const int64_t n = dataset->GetRows();
        auto out = std::make_unique<float[]>(std::max<int64_t>(n * dim_, 1));
        std::shared_lock<std::shared_mutex> lk(rw_);
        if (sh_.size() == 1) {
            const int rc = knhip_index_get_vectors(sh_[0].idx.p, n, dataset->GetIds(), out.get());
            if (rc) return expected<DataSetPtr>::Err(ToStatus(rc), knhip_last_error());
        } else {
            // every shard fills the rows it stores; an id stored nowhere is an error, as on one device
            std::vector<uint8_t> found((size_t)n, 0), f((size_t)n);
            for (const auto& s : sh_) {
                const int rc = knhip_index_find_vectors(s.idx.p, n, dataset->GetIds(), out.get(), f.data());
                if (rc) return expected<DataSetPtr>::Err(ToStatus(rc), knhip_last_error());
                for (int64_t i = 0; i < n; i++) found[(size_t)i] |= f[(size_t)i];
            }
            for (int64_t i = 0; i < n; i++) {
                if (!found[(size_t)i]) return expected<DataSetPtr>::Err(Status::invalid_args, "get_vectors: id not in the index");
            }
        }
        return GenResultDataSet(n, dim_, (const void*)out.release());
    }

    static bool
    StaticHasRawData(const knowhere::BaseConfig& config, const IndexVersion& /*version*/) {
        // cosine stores normalised vectors (IvfIndexNode::StaticHasRawData semantics, ivf.cc:142-160)
        const bool cosine = config.metric_type.has_value() && IsMetricType(config.metric_type.value(), metric::COSINE);
        return (Kind == KNHIP_BRUTE_FORCE || Kind == KNHIP_IVF_FLAT) && !cosine;
    }
    bool
    HasRawData(const std::string& metric_type) const override {
        return (Kind == KNHIP_BRUTE_FORCE || Kind == KNHIP_IVF_FLAT) && !IsMetricType(metric_type, metric::COSINE);
    }

    expected<DataSetPtr>
    GetIndexMeta(std::unique_ptr<Config> /*cfg*/) const override {
        return expected<DataSetPtr>::Err(Status::not_implemented, "GetIndexMeta not implemented");
    }

    // One named blob in the BinarySet (ivf.cc:1717-1744), holding the FAISS byte format the CPU nodes write
    // (faiss_io.h): IxF2/IxFI (IxF9 for cosine), IwFl, IwSq, IwPQ, wrapped in IxRF when built with refine.  The trained
    // state and the inverted lists are read back from the device(s); a sharded index writes the SAME bytes as a
    // single-device one (every list comes from the device that owns it).
    Status
    Serialize(BinarySet& binset) const override {
        if (!Built()) return Status::empty_index;
        using namespace knhip_host;
        std::shared_lock<std::shared_mutex> lk(rw_);
        const int64_t count = CountLocked();
        FaissIndexData x;
        auto fill_hdr = [&](FaissHeader& h, int64_t ntotal, bool cosine_byte) {
            h.d = (int32_t)dim_;
            h.ntotal = ntotal;
            h.dummy[0] = cosine_byte ? 1 : 0;  // Knowhere's is_cosine byte (cppcontrib/knowhere/impl/index_write.cpp:84-88)
            h.is_trained = true;
            h.metric = metric_ == KNHIP_L2 ? 1 : 0;
        };
        const uint32_t flat_cc = metric_ == KNHIP_L2 ? FourCC("IxF2") : FourCC("IxFI");
        fill_hdr(x.hdr, count, cosine_);
        int rc = 0;
        // rows kept in per-shard id ranges (FLAT base, refine store) -> one array in id order
        auto read_rows = [&](KnhipHandle Shard::*which, std::vector<float>* out) -> int {
            out->resize((size_t)count * dim_);
            size_t pos = 0;
            for (const auto& s : sh_) {
                const knhip_index* h = (s.*which).p;
                const int64_t n = h ? knhip_index_count(h) : 0;
                if (n == 0) continue;
                if (pos + (size_t)n * dim_ > out->size()) return KNHIP_ERR_INVALID_ARGS;
                if (int e = knhip_index_get_lists(h, (uint8_t*)(out->data() + pos), nullptr)) return e;
                pos += (size_t)n * dim_;
            }
            return pos == out->size() ? KNHIP_OK : KNHIP_ERR_INVALID_ARGS;
        };
        if constexpr (Kind == KNHIP_BRUTE_FORCE) {
            if ((rc = read_rows(&Shard::idx, &x.xb))) return ToStatus(rc);
            if (cosine_) {  // IndexFlatCosine: "IxF9" = header, raw rows, L2 norms = 1 / inverse norm
                x.fourcc = FourCC("IxF9");  // (index_write.cpp:539-546, L2NormsStorage::as_l2_norms IndexCosine.cpp:261-268)
                x.flat_norms.resize((size_t)count);
                for (int64_t i = 0; i < count; i++) x.flat_norms[(size_t)i] = 1.0f / row_scale_by_id_[(size_t)i];
            } else {
                x.fourcc = flat_cc;
            }
        } else {
            x.fourcc = Kind == KNHIP_IVF_FLAT ? FourCC("IwFl") : Kind == KNHIP_IVF_PQ ? FourCC("IwPQ") : FourCC("IwSq");
            x.nlist = (uint64_t)nlist_;
            x.nprobe = (uint64_t)default_nprobe_;
            x.quantizer.fourcc = flat_cc;
            fill_hdr(x.quantizer.hdr, nlist_, false);
            x.quantizer.xb.resize((size_t)nlist_ * dim_);
            const knhip_index* first = sh_[0].idx.p;  // (the trained state is the same on every shard)
            if ((rc = knhip_index_get_coarse(first, x.quantizer.xb.data()))) return ToStatus(rc);
            x.by_residual = true;
            x.code_size = (uint64_t)CodeSize();
            if constexpr (Kind == KNHIP_IVF_PQ) {
                x.pq_d = (uint64_t)dim_;
                x.pq_M = (uint64_t)m_;
                x.pq_nbits = (uint64_t)nbits_;
                x.pq_centroids.resize(((size_t)1 << nbits_) * dim_);
                if ((rc = knhip_index_get_pq(first, x.pq_centroids.data()))) return ToStatus(rc);
            } else if constexpr (Kind == KNHIP_IVF_SQ8) {
                x.sq_qtype = 0;      // ScalarQuantizer::QT_8bit
                x.sq_rangestat = 0;  // RS_minmax
                x.sq_d = (uint64_t)dim_;
                x.sq_code_size = (uint64_t)dim_;
                x.sq_trained.resize((size_t)2 * dim_);
                if ((rc = knhip_index_get_sq(first, x.sq_trained.data(), x.sq_trained.data() + dim_))) return ToStatus(rc);
            }
            x.codes.assign(nlist_, {});
            x.ids.assign(nlist_, {});
            std::vector<int64_t> all_sizes((size_t)nlist_, 0);
            for (const auto& s : sh_) {
                const int64_t cnt = knhip_index_count(s.idx.p);
                if (cnt == 0) continue;
                std::vector<int64_t> sizes((size_t)nlist_);
                if ((rc = knhip_index_get_list_sizes(s.idx.p, sizes.data()))) return ToStatus(rc);
                std::vector<uint8_t> codes((size_t)cnt * CodeSize());
                std::vector<int64_t> ids((size_t)cnt);
                if ((rc = knhip_index_get_lists(s.idx.p, codes.data(), ids.data()))) return ToStatus(rc);
                int64_t pos = 0;
                for (int64_t l = 0; l < nlist_; l++) {
                    if (sizes[l] == 0) continue;
                    if (all_sizes[l] != 0) return Status::faiss_inner_error;  // (a list lives on exactly one shard)
                    x.codes[l].assign(codes.begin() + pos * CodeSize(), codes.begin() + (pos + sizes[l]) * CodeSize());
                    x.ids[l].assign(ids.begin() + pos, ids.begin() + pos + sizes[l]);
                    all_sizes[l] = sizes[l];
                    pos += sizes[l];
                }
            }
            size_t non0 = 0;
            for (int64_t l = 0; l < nlist_; l++) non0 += all_sizes[l] != 0;
            x.lists_sparse = !(non0 > (size_t)nlist_ / 2);  // index_write.cpp:309-316
            if (Kind == KNHIP_IVF_FLAT && cosine_) {        // Knowhere cosine IVF-Flat carries the row norms
                x.with_norm = true;
                x.norms.assign(nlist_, {});
                for (int64_t l = 0; l < nlist_; l++) {
                    x.norms[l].resize((size_t)all_sizes[l]);
                    for (int64_t j = 0; j < all_sizes[l]; j++) x.norms[l][(size_t)j] = row_scale_by_id_[(size_t)x.ids[l][(size_t)j]];
                }
            }
            if (has_refine_ && sh_[0].rows.p) {  // IndexRefine over an IndexScalarQuantizer ("IxSQ", refine_utils.cc:150-185)
                const knhip_rows* rs = sh_[0].rows.p;
                int64_t have = 0;
                for (const auto& sd : sh_) have += sd.rows.p ? knhip_rows_count(sd.rows.p) : 0;
                if (have != count) return Status::invalid_index_error;
                x.has_refine = x.refine_is_sq = true;
                fill_hdr(x.refine_hdr, count, cosine_);
                FaissSQFlat& sq = x.refine_sq;
                fill_hdr(sq.hdr, count, false);
                // (ScalarQuantizer::QuantizerType, impl/ScalarQuantizer.h:27-40: QT_8bit 0, QT_fp16 4, QT_6bit 6, QT_bf16 7,
                // QT_8bit_direct_signed 8)
                sq.qtype = refine_rows_type_ == KNHIP_ROWS_FP16   ? 4
                           : refine_rows_type_ == KNHIP_ROWS_BF16 ? 7
                           : refine_rows_type_ == KNHIP_ROWS_SQ6  ? 6
                           : refine_rows_type_ == KNHIP_ROWS_INT8 ? 8
                           : refine_rows_type_ == KNHIP_ROWS_SQ4U ? 3
                                                                  : 0;
                sq.rangestat = 0;  // RS_minmax, rangestat_arg 0: the ScalarQuantizer defaults
                if (refine_rows_type_ == KNHIP_ROWS_SQ4U && metric_ == KNHIP_L2) {
                    sq.rangestat = 2;  // RS_quantiles, as Knowhere sets it for QT_4bit_uniform + L2 (the fields are written as set)
                    sq.rangestat_arg = 0.01f;
                }
                sq.d = (uint64_t)dim_;
                sq.code_size = (uint64_t)knhip_rows_code_size(rs);
                if (RowsRanged()) {
                    const size_t nr = RowsRangeLen();
                    sq.trained.resize(2 * nr);
                    if ((rc = knhip_rows_get_trained(rs, sq.trained.data(), sq.trained.data() + nr))) return ToStatus(rc);
                }
                sq.codes.resize((size_t)count * sq.code_size);
                for (const auto& sd : sh_) {  // (id ranges in shard order: raw_base ascending)
                    if (!sd.rows.p || knhip_rows_count(sd.rows.p) == 0) continue;
                    if (sd.raw_base < 0 || sd.raw_base + knhip_rows_count(sd.rows.p) > count) return Status::invalid_index_error;
                    if ((rc = knhip_rows_get_codes(sd.rows.p, sq.codes.data() + (size_t)sd.raw_base * sq.code_size)))
                        return ToStatus(rc);
                }
                x.k_factor = 1.f;
            } else if (has_refine_ && sh_[0].raw.p) {  // IndexRefineFlat (ivf.cc:673-700)
                x.has_refine = true;
                fill_hdr(x.refine_hdr, count, cosine_);
                x.refine_index.fourcc = flat_cc;
                fill_hdr(x.refine_index.hdr, count, false);
                if ((rc = read_rows(&Shard::raw, &x.refine_index.xb))) return ToStatus(rc);
                x.k_factor = 1.f;
            }
        }
        std::vector<uint8_t> buf;
        std::string err;
        if (!WriteFaissIndex(x, &buf, &err)) return Status::faiss_inner_error;
        std::shared_ptr<uint8_t[]> data(new uint8_t[buf.size()]);
        std::memcpy(data.get(), buf.data(), buf.size());
        binset.Append(Type(), data, (int64_t)buf.size());
        return Status::success;
    }

    // Accepts the blob of this node AND of the CPU node of the same kind (FLAT / IVF_FLAT / IVF_PQ / IVF_SQ8, or the
    // knowhere-1.x name "IVF", ivf.cc:1750-1757): same bytes.  Every length and cross-field relation is checked before a
    // pointer is handed to the C ABI: a corrupted BinarySet yields invalid_serialized_index_type, never a wild read.
    Status
    Deserialize(const BinarySet& binset, std::shared_ptr<Config> cfg) override {
        using namespace knhip_host;
        static const char* cpu_names[] = {"FLAT", "IVF_FLAT", "IVF_PQ", "IVF_SQ8"};
        BinaryPtr b = binset.GetByName(Type());
        if (!b) b = binset.GetByName(cpu_names[Kind]);
        if (!b && Kind != KNHIP_BRUTE_FORCE) b = binset.GetByName("IVF");
        if (!b) return Status::invalid_binary_set;
        if (knhip_abi_version() != KNHIP_ABI_VERSION) return Status::cuda_runtime_error;
        FaissIndexData x;
        std::string err;
        if (!ParseFaissIndex(b->data.get(), (size_t)b->size, &x, &err)) return Status::invalid_serialized_index_type;
        const bool flat = x.fourcc == FourCC("IxF2") || x.fourcc == FourCC("IxFI") || x.fourcc == FourCC("IxF9");
        const bool kind_ok = (Kind == KNHIP_BRUTE_FORCE && flat) || (Kind == KNHIP_IVF_FLAT && x.fourcc == FourCC("IwFl")) ||
                             (Kind == KNHIP_IVF_PQ && x.fourcc == FourCC("IwPQ")) ||
                             (Kind == KNHIP_IVF_SQ8 && x.fourcc == FourCC("IwSq"));
        if (!kind_ok) return Status::invalid_serialized_index_type;
        if (x.hdr.metric != 0 && x.hdr.metric != 1) return Status::invalid_metric_type;
        const int64_t d = x.hdr.d, ntotal = x.hdr.ntotal;
        if (d <= 0 || d > 65536 || ntotal < 0) return Status::invalid_serialized_index_type;
        if (flat) {
            if (x.xb.size() != (size_t)ntotal * d) return Status::invalid_serialized_index_type;
            if (!x.flat_norms.empty() && x.flat_norms.size() != (size_t)ntotal) return Status::invalid_serialized_index_type;
        } else {
            if (x.quantizer.hdr.metric != x.hdr.metric) return Status::not_implemented;
            if (x.nlist == 0 || x.nlist > 65536 * 16 || x.quantizer.hdr.d != d ||
                x.quantizer.xb.size() != (size_t)x.nlist * d || x.codes.size() != x.nlist || x.ids.size() != x.nlist)
                return Status::invalid_serialized_index_type;
            if (Kind == KNHIP_IVF_PQ && (x.pq_nbits < 1 || x.pq_nbits > 8 || !x.by_residual || x.pq_M < 1 || x.pq_M > 128 ||
                                         (uint64_t)d / x.pq_M > 144))
                return Status::not_implemented;
            // (PQ codes on the wire: M indices of nbits bits as a little-endian bit string, ProductQuantizer.cpp:69)
            const uint64_t want_cs = Kind == KNHIP_IVF_FLAT ? (uint64_t)d * 4
                                     : Kind == KNHIP_IVF_PQ ? (x.pq_M * x.pq_nbits + 7) / 8 : (uint64_t)d;
            if (Kind == KNHIP_IVF_PQ && (x.pq_d != (uint64_t)d || d % (int64_t)x.pq_M != 0 ||
                                          x.pq_centroids.size() != ((size_t)1 << x.pq_nbits) * d))
                return Status::invalid_serialized_index_type;
            if (Kind == KNHIP_IVF_SQ8 && (x.sq_qtype != 0 || !x.by_residual)) return Status::not_implemented;
            if (Kind == KNHIP_IVF_SQ8 && (x.sq_d != (uint64_t)d || x.sq_code_size != (uint64_t)d ||
                                           x.sq_trained.size() != 2 * (size_t)d))
                return Status::invalid_serialized_index_type;
            if (Kind != KNHIP_IVF_FLAT && x.code_size != want_cs) return Status::invalid_serialized_index_type;
            int64_t sum = 0;
            for (uint64_t l = 0; l < x.nlist; l++) {
                const size_t n = x.ids[l].size();
                if (x.codes[l].size() != n * want_cs) return Status::invalid_serialized_index_type;
                if (x.with_norm && (x.norms.size() != x.nlist || x.norms[l].size() != n))
                    return Status::invalid_serialized_index_type;
                sum += (int64_t)n;
            }
            if (sum != ntotal) return Status::invalid_serialized_index_type;
            if (x.has_refine) {
                if (x.refine_is_sq ? (x.refine_sq.hdr.ntotal != ntotal || x.refine_sq.hdr.d != d)
                                   : (x.refine_index.hdr.ntotal != ntotal || x.refine_index.hdr.d != d ||
                                      x.refine_index.xb.size() != (size_t)ntotal * d))
                    return Status::invalid_serialized_index_type;
                for (uint64_t l = 0; l < x.nlist; l++) {
                    for (int64_t id : x.ids[l]) {
                        if (id < 0 || id >= ntotal) return Status::invalid_serialized_index_type;
                    }
                }
            }
        }
        const int new_metric = x.hdr.metric == 1 ? KNHIP_L2 : KNHIP_IP;
        bool new_cosine = x.hdr.is_cosine() || x.fourcc == FourCC("IxF9") || x.with_norm;
        std::vector<int32_t> devs;
        if (cfg) {
            const auto& c = static_cast<const knowhere_config_type&>(*cfg);
            if (c.metric_type.has_value() && IsMetricType(c.metric_type.value(), metric::COSINE) && new_metric == KNHIP_IP)
                new_cosine = true;
            if (Status st = SelectDevices(c, /*deserialize=*/true, &devs); st != Status::success) return st;
        } else {
            knowhere_config_type none;
            if (Status st = SelectDevices(none, /*deserialize=*/true, &devs); st != Status::success) return st;
        }
        // The CPU cosine indexes keep the RAW rows plus their L2 norms (FLAT: the wire carries the norms, the index
        // multiplies by their inverses, L2NormsStorage::add_l2_norms IndexCosine.cpp:247-255; IVF_FLAT: ip / norm,
        // cppcontrib/knowhere/IndexIVFFlat.cpp:199-210): kept exactly so, per row id
        std::vector<float> scale_by_id;
        if (new_cosine && flat) {
            if (x.flat_norms.size() != (size_t)ntotal) return Status::invalid_serialized_index_type;
            scale_by_id.resize((size_t)ntotal);
            for (int64_t i = 0; i < ntotal; i++) {
                const float nr = x.flat_norms[(size_t)i];
                scale_by_id[(size_t)i] = nr == 0.0f ? 1.0f : (1.0f / nr);
            }
        }
        if (new_cosine && Kind == KNHIP_IVF_FLAT) {
            if (!x.with_norm) return Status::invalid_serialized_index_type;
            int64_t max_id = -1;
            for (uint64_t l = 0; l < x.nlist; l++) {
                if (x.norms[l].size() != x.ids[l].size()) return Status::invalid_serialized_index_type;
                for (int64_t id : x.ids[l]) {
                    if (id < 0) return Status::invalid_serialized_index_type;
                    max_id = std::max(max_id, id);
                }
            }
            if (max_id >= 4 * std::max<int64_t>(ntotal, 1) + 1024) return Status::invalid_serialized_index_type;
            scale_by_id.assign((size_t)(max_id + 1), 1.0f);
            for (uint64_t l = 0; l < x.nlist; l++) {
                for (size_t j = 0; j < x.ids[l].size(); j++) scale_by_id[(size_t)x.ids[l][j]] = x.norms[l][j];
            }
        }
        std::unique_lock<std::shared_mutex> lk(rw_);
        DropShards();
        metric_ = new_metric;
        cosine_ = new_cosine;
        metric_name_ = cosine_ ? metric::COSINE : (metric_ == KNHIP_L2 ? metric::L2 : metric::IP);
        dim_ = d;
        nlist_ = (int64_t)x.nlist;
        if (x.nprobe >= 1 && x.nprobe <= 65536) default_nprobe_ = (int64_t)x.nprobe;  // the index's default nprobe
        m_ = (int64_t)x.pq_M;
        nbits_ = Kind == KNHIP_IVF_PQ ? (int64_t)x.pq_nbits : 8;
        has_refine_ = x.has_refine;
        refine_rows_type_ = !(x.has_refine && x.refine_is_sq) ? 0
                            : x.refine_sq.qtype == 4          ? KNHIP_ROWS_FP16
                            : x.refine_sq.qtype == 7          ? KNHIP_ROWS_BF16
                            : x.refine_sq.qtype == 6          ? KNHIP_ROWS_SQ6
                            : x.refine_sq.qtype == 8          ? KNHIP_ROWS_INT8
                            : x.refine_sq.qtype == 3          ? KNHIP_ROWS_SQ4U
                                                              : KNHIP_ROWS_SQ8;
        row_scale_by_id_ = std::move(scale_by_id);
        if (Status st = CreateShards(devs); st != Status::success) return st;
        const int W = (int)sh_.size();
        int rc = KNHIP_OK;
        auto bail = [&](int e) {
            DropShards();
            return ToStatus(e);
        };
        auto bail_st = [&](Status st) { // (a post-step failed: the half-loaded index is dropped, as for every other step)
            DropShards();
            return st;
        };
        if constexpr (Kind == KNHIP_BRUTE_FORCE) {
            if ((rc = AddRowRanges(&Shard::idx, &Shard::row_base, 0, ntotal, x.xb.data()))) return bail(rc);
            if (StoredNormCosine()) {
                if (Status st = PushRowScale(); st != Status::success) return bail_st(st);
            }
            return Status::success;
        }
        std::vector<int64_t> sizes(nlist_);
        for (int64_t l = 0; l < nlist_; l++) sizes[l] = (int64_t)x.ids[l].size();
        if (W > 1) owner_ = DealLists(sizes, W);
        for (int r = 0; r < W; r++) {
            knhip_index* h = sh_[r].idx.p;
            if ((rc = knhip_index_set_coarse(h, x.quantizer.xb.data()))) return bail(rc);
            if (Kind == KNHIP_IVF_PQ && (rc = knhip_index_set_pq(h, x.pq_centroids.data()))) return bail(rc);
            if (Kind == KNHIP_IVF_SQ8 && (rc = knhip_index_set_sq(h, x.sq_trained.data(), x.sq_trained.data() + dim_)))
                return bail(rc);
            std::vector<int64_t> sz(nlist_, 0);
            std::vector<const uint8_t*> cp(nlist_, nullptr);
            std::vector<const int64_t*> ip(nlist_, nullptr);
            for (int64_t l = 0; l < nlist_; l++) {
                if (W > 1 && owner_[(size_t)l] != r) continue;
                sz[l] = sizes[l];
                cp[l] = x.codes[l].data();
                ip[l] = x.ids[l].data();
            }
            if ((rc = knhip_index_add_lists(h, sz.data(), cp.data(), ip.data()))) return bail(rc);
        }
        if (StoredNormCosine()) {
            const Status st = PushRowScale();
            if (st != Status::success) return bail_st(st);
        }
        if (x.has_refine && x.refine_is_sq) {
            const FaissSQFlat& sq = x.refine_sq;
            if ((rc = CreateRowStores())) return bail(rc);
            for (int r = 0; r < W; r++) {
                if (RowsRanged() &&
                    (rc = knhip_rows_set_trained(sh_[r].rows.p, sq.trained.data(), sq.trained.data() + RowsRangeLen())))
                    return bail(rc);
                const int64_t lo = ntotal * r / W, hi = ntotal * (r + 1) / W;
                sh_[r].raw_base = lo;
                if (hi > lo && (rc = knhip_rows_add_codes(sh_[r].rows.p, hi - lo, sq.codes.data() + (size_t)lo * sq.code_size)))
                    return bail(rc);
            }
            if (W > 1) {
                if (Status st = AttachRawToGroup(); st != Status::success) return bail_st(st);
            }
            return Status::success;
        }
        // raw rows for refine, in id order
        if (x.has_refine && !x.refine_index.xb.empty()) {
            for (auto& s : sh_) {
                knhip_desc rd{};
                rd.kind = KNHIP_BRUTE_FORCE;
                rd.metric = metric_;
                rd.dim = (int32_t)dim_;
                rd.device = s.device;
                if ((rc = knhip_index_create(&rd, &s.raw.p))) return bail(rc);
            }
            if ((rc = AddRowRanges(&Shard::raw, &Shard::raw_base, 0, ntotal, x.refine_index.xb.data()))) return bail(rc);
            if (W > 1) {
                if (Status st = AttachRawToGroup(); st != Status::success) return bail_st(st);
            }
        }
        return Status::success;
    }

    Status
    // IvfIndexNode::DeserializeFromFile (ivf.cc:1838-1916) reads the faiss index bytes from a file
    // (faiss::read_index(filename, io_flags); enable_mmap only changes how the CPU node keeps them): the file holds
    // exactly the blob Serialize puts into the BinarySet, so it is read once and handed to Deserialize -- the lists go
    // to HBM either way.  (The cuVS node returns not_implemented here, gpu_cuvs.h:250-253.)
    DeserializeFromFile(const std::string& filename, std::shared_ptr<Config> config) override {
        FILE* f = std::fopen(filename.c_str(), "rb");
        if (f == nullptr) return Status::disk_file_error;
        std::fseek(f, 0, SEEK_END);
        const long size = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        if (size <= 0) {
            std::fclose(f);
            return Status::invalid_serialized_index_type;
        }
        std::shared_ptr<uint8_t[]> data(new uint8_t[(size_t)size]);
        const size_t got = std::fread(data.get(), 1, (size_t)size, f);
        std::fclose(f);
        if (got != (size_t)size) return Status::disk_file_error;
        BinarySet binset;
        binset.Append(Type(), data, (int64_t)size);
        return Deserialize(binset, std::move(config));
    }

    static std::unique_ptr<BaseConfig>
    StaticCreateConfig() {
        return std::make_unique<knowhere_config_type>();
    }
    std::unique_ptr<BaseConfig>
    CreateConfig() const override {
        return StaticCreateConfig();
    }
    static Status
    StaticConfigCheck(const knowhere::BaseConfig& config, PARAM_TYPE paramType, std::string& msg) {
        // what the typed config cannot express: this backend needs at least one visible device, and a device list that
        // names devices that exist
        if (paramType == PARAM_TYPE::TRAIN || paramType == PARAM_TYPE::DESERIALIZE) {
            const int ndev = knhip_device_count();
            if (ndev <= 0) {
                msg = "no HIP device available for " + std::string(TypeName());
                return Status::cuda_runtime_error;
            }
            const auto* c = dynamic_cast<const knowhere_config_type*>(&config);
            if (c && c->gpu_ids.has_value() && HipParseGpuIds(c->gpu_ids.value(), ndev).empty()) {
                msg = "gpu_ids \"" + c->gpu_ids.value() + "\": comma separated device ordinals below " + std::to_string(ndev) +
                      ", or \"all\"";
                return Status::invalid_args;
            }
            if (c && c->gpu_id.has_value() && c->gpu_id.value() >= ndev) {
                msg = "gpu_id " + std::to_string(c->gpu_id.value()) + ": " + std::to_string(ndev) + " device(s) visible";
                return Status::invalid_args;
            }
        }
        return Status::success;
    }

    int64_t
    Dim() const override {
        return dim_;
    }
    int64_t
    Size() const override {
        int64_t b = 0;
        for (const auto& s : sh_) {
            b += (s.idx.p ? knhip_index_device_bytes(s.idx.p) : 0) + (s.raw.p ? knhip_index_device_bytes(s.raw.p) : 0) +
                 (s.rows.p ? knhip_rows_device_bytes(s.rows.p) : 0);
        }
        return b;
    }
    int64_t
    Count() const override {
        return CountLocked();
    }
    static const char*
    TypeName() {
        switch (Kind) {
            case KNHIP_BRUTE_FORCE: return IndexEnum::INDEX_HIP_BRUTEFORCE;
            case KNHIP_IVF_FLAT: return IndexEnum::INDEX_HIP_IVFFLAT;
            case KNHIP_IVF_PQ: return IndexEnum::INDEX_HIP_IVFPQ;
            default: return IndexEnum::INDEX_HIP_IVFSQ8;
        }
    }
    std::string
    Type() const override {
        return TypeName();
    }

    // the devices this index lives on, in shard order (tests; Milvus would log it)
    std::vector<int32_t>
    Devices() const {
        std::vector<int32_t> d;
        for (const auto& s : sh_) d.push_back(s.device);
        return d;
    }

 private:
    bool
    Built() const {
        return !sh_.empty() && sh_[0].idx.p != nullptr;
    }
    int64_t
    CountLocked() const {
        int64_t n = 0;
        for (const auto& s : sh_) n += s.idx.p ? knhip_index_count(s.idx.p) : 0;
        return n;
    }
    int64_t
    CodeSize() const {
        return Kind == KNHIP_IVF_FLAT ? dim_ * 4 : (Kind == KNHIP_IVF_PQ ? (m_ * nbits_ + 7) / 8 : dim_);
    }
    // a second device-resident store of the raw rows: IndexRefineFlat only (GetVectorByIds of IVF_FLAT is served from the
    // index's own rows through knhip_index_get_vectors' direct map)
    bool
    NeedRawStore() const {
        return (Kind == KNHIP_IVF_PQ || Kind == KNHIP_IVF_SQ8) && has_refine_;
    }
    // COSINE on FLAT / IVF_FLAT: raw rows + one float per row (inverse norm / norm), see Add
    bool
    StoredNormCosine() const {
        return cosine_ && (Kind == KNHIP_BRUTE_FORCE || Kind == KNHIP_IVF_FLAT);
    }

    static bool
    MaterialiseBitset(const BitsetView& bitset, std::vector<uint8_t>* store, const uint8_t** bits, int64_t* nbits) {
        const bool has_bitset = bitset.data() != nullptr && bitset.num_bits() != 0;
        if (has_bitset && bitset.has_out_ids()) {
            const size_t n_in = bitset.out_ids_count();
            store->assign((n_in + 7) / 8, 0);
            for (size_t i = 0; i < n_in; i++) {
                if (bitset.test((int64_t)i)) (*store)[i >> 3] |= (uint8_t)(1u << (i & 7));
            }
            *bits = store->data();
            *nbits = (int64_t)n_in;
        } else if (has_bitset) {
            *bits = bitset.data();
            *nbits = (int64_t)bitset.num_bits();
        }
        return has_bitset;
    }

    // ---- devices ---------------------------------------------------------------------------------------------------
    // gpu_ids (several: sharded) > gpu_id > the reference's placement rule
    static Status
    SelectDevices(const knowhere_config_type& c, bool deserialize, std::vector<int32_t>* out) {
        const int ndev = knhip_device_count();
        if (ndev <= 0) return Status::cuda_runtime_error;
        out->clear();
        if (c.gpu_ids.has_value()) {
            *out = HipParseGpuIds(c.gpu_ids.value(), ndev);
            if (out->empty()) {
                LOG_KNOWHERE_ERROR_ << TypeName() << ": gpu_ids \"" << c.gpu_ids.value() << "\" names no valid device list ("
                                    << ndev << " visible)";
                return Status::invalid_args;
            }
            if (out->size() > 64) return Status::invalid_args;
            return Status::success;
        }
        if (c.gpu_id.has_value()) {
            if (c.gpu_id.value() < 0 || c.gpu_id.value() >= ndev) return Status::invalid_args;
            out->push_back(c.gpu_id.value());
            return Status::success;
        }
        out->push_back(deserialize ? MostFreeDevice(ndev) : RoundRobinDevice(ndev));
        return Status::success;
    }

    void
    DropShards() {
        group_.reset();
        sh_.clear();
        owner_.clear();
    }

    // one knhip_index per device (empty, untrained) + the shard group when there are several
    Status
    CreateShards(const std::vector<int32_t>& devs) {
        DropShards();
        sh_.resize(devs.size());
        for (size_t r = 0; r < devs.size(); r++) {
            sh_[r].device = devs[r];
            knhip_desc desc{};
            desc.kind = Kind;
            desc.metric = metric_;
            desc.dim = (int32_t)dim_;
            desc.device = devs[r];
            if constexpr (Kind != KNHIP_BRUTE_FORCE) desc.nlist = nlist_;
            if constexpr (Kind == KNHIP_IVF_PQ) {
                desc.pq_m = (int32_t)m_;
                desc.pq_nbits = (int32_t)nbits_;
            }
            if (int rc = knhip_index_create(&desc, &sh_[r].idx.p)) {
                DropShards();
                return ToStatus(rc);
            }
        }
        g_last_placement = devs;
        if (devs.size() > 1) {
            bool distinct = true;
            for (size_t a = 0; a < devs.size(); a++) {
                for (size_t b = a + 1; b < devs.size(); b++) distinct = distinct && devs[a] != devs[b];
            }
            // RCCL over xGMI between distinct devices; shards sharing a device exchange by device copies
            const int rc = knhip_shard_group_create((int32_t)devs.size(), devs.data(),
                                                    distinct ? KNHIP_SHARDS_RCCL : KNHIP_SHARDS_STAGED, &group_.p);
            if (rc) {
                LOG_KNOWHERE_ERROR_ << TypeName() << ": shard group: " << knhip_shard_group_last_error();
                DropShards();
                return ToStatus(rc);
            }
            for (size_t r = 0; r < devs.size(); r++) {
                if (int e = knhip_shard_group_set_index(group_.p, (int32_t)r, sh_[r].idx.p)) {
                    DropShards();
                    return ToStatus(e);
                }
            }
        }
        return Status::success;
    }

    // centroids / codebooks / SQ ranges of shard 0 -> every other shard (every device quantises identically)
    int
    ReplicateTrainedState() {
        std::vector<float> cen((size_t)nlist_ * dim_), aux;
        if (int rc = knhip_index_get_coarse(sh_[0].idx.p, cen.data())) return rc;
        if constexpr (Kind == KNHIP_IVF_PQ) {
            aux.resize(((size_t)1 << nbits_) * dim_);
            if (int rc = knhip_index_get_pq(sh_[0].idx.p, aux.data())) return rc;
        } else if constexpr (Kind == KNHIP_IVF_SQ8) {
            aux.resize((size_t)2 * dim_);
            if (int rc = knhip_index_get_sq(sh_[0].idx.p, aux.data(), aux.data() + dim_)) return rc;
        }
        for (size_t r = 1; r < sh_.size(); r++) {
            knhip_index* h = sh_[r].idx.p;
            if (int rc = knhip_index_set_coarse(h, cen.data())) return rc;
            if constexpr (Kind == KNHIP_IVF_PQ) {
                if (int rc = knhip_index_set_pq(h, aux.data())) return rc;
            } else if constexpr (Kind == KNHIP_IVF_SQ8) {
                if (int rc = knhip_index_set_sq(h, aux.data(), aux.data() + dim_)) return rc;
            }
        }
        return KNHIP_OK;
    }

    // rows kept by id range (the FLAT base, the refine store): the first batch is cut into one contiguous range per
    // shard; later batches extend the LAST shard's range (a BRUTE_FORCE knhip_index is one id range: row + offset)
    int
    AddRowRanges(KnhipHandle Shard::*which, int64_t Shard::*base, int64_t n0, int64_t rows, const float* x) {
        const int W = (int)sh_.size();
        if (n0 == 0) {
            if (rows < W && which == &Shard::idx) return KNHIP_ERR_INVALID_ARGS;  // (fewer rows than devices)
            for (int r = 0; r < W; r++) {
                const int64_t lo = rows * r / W, hi = rows * (r + 1) / W;
                if (hi == lo) continue;  // (an empty range of the refine store: the group skips it)
                sh_[r].*base = lo;
                if (int rc = knhip_index_add_vectors((sh_[r].*which).p, hi - lo, x + lo * dim_, nullptr, lo)) return rc;
            }
            return KNHIP_OK;
        }
        return knhip_index_add((sh_[W - 1].*which).p, rows, x, nullptr);
    }

    // one quantised refine store per device
    int
    CreateRowStores() {
        for (auto& sd : sh_) {
            if (int rc = knhip_rows_create(sd.device, (int32_t)dim_, refine_rows_type_, &sd.rows.p)) return rc;
        }
        return KNHIP_OK;
    }
    // refine stores with trained ranges: per dimension (sq8, sq6) or one for all dimensions (sq4u)
    bool
    RowsRanged() const {
        return refine_rows_type_ == KNHIP_ROWS_SQ8 || refine_rows_type_ == KNHIP_ROWS_SQ6 || refine_rows_type_ == KNHIP_ROWS_SQ4U;
    }
    size_t
    RowsRangeLen() const {  // floats per half (vmin | vdiff) of the trained vector
        return refine_rows_type_ == KNHIP_ROWS_SQ4U ? 1 : (size_t)dim_;
    }
    // the ranges trained on the first device -> every other store (all devices decode alike)
    int
    ReplicateRowRanges() {
        if (!RowsRanged() || sh_.size() < 2) return KNHIP_OK;
        const size_t nr = RowsRangeLen();
        std::vector<float> tr(2 * nr);
        if (int rc = knhip_rows_get_trained(sh_[0].rows.p, tr.data(), tr.data() + nr)) return rc;
        for (size_t r = 1; r < sh_.size(); r++) {
            if (int rc = knhip_rows_set_trained(sh_[r].rows.p, tr.data(), tr.data() + nr)) return rc;
        }
        return KNHIP_OK;
    }

    Status
    AttachRawToGroup() {
        if (refine_rows_type_ != 0) {
            for (size_t r = 0; r < sh_.size(); r++) {
                const bool any = sh_[r].rows.p && knhip_rows_count(sh_[r].rows.p) > 0;
                if (int rc = knhip_shard_group_set_raw_rows(group_.p, (int32_t)r, any ? sh_[r].rows.p : nullptr, sh_[r].raw_base))
                    return ToStatus(rc);
            }
            return Status::success;
        }
        for (size_t r = 0; r < sh_.size(); r++) {
            const float* d_rows = nullptr;
            const int64_t n = sh_[r].raw.p ? knhip_index_count(sh_[r].raw.p) : 0;
            if (n > 0) {
                if (int rc = knhip_index_get_vectors_device(sh_[r].raw.p, &d_rows)) return ToStatus(rc);
            }
            if (int rc = knhip_shard_group_set_raw(group_.p, (int32_t)r, d_rows, n, sh_[r].raw_base)) return ToStatus(rc);
        }
        return Status::success;
    }

    // IVF kinds over several devices: quantizer->assign once (knhip_index_assign on the first device), the first batch
    // fixes the owner of every list (size-balanced deal), every row goes to the device that owns its list with its id
    // (the running row number), the devices encode and append their rows concurrently.
    int
    AddSharded(int64_t n0, int64_t rows, const float* x_store, const float* x_assign) {
        const int W = (int)sh_.size();
        std::vector<int64_t> assign((size_t)rows);
        if (int rc = knhip_index_assign(sh_[0].idx.p, rows, x_assign, assign.data())) return rc;
        if (owner_.empty()) {
            std::vector<int64_t> sizes((size_t)nlist_, 0);
            for (int64_t i = 0; i < rows; i++) {
                if (assign[(size_t)i] >= 0 && assign[(size_t)i] < nlist_) sizes[(size_t)assign[(size_t)i]]++;
            }
            owner_ = DealLists(sizes, W);
        }
        std::vector<std::vector<float>> xs((size_t)W), xa((size_t)W);
        std::vector<std::vector<int64_t>> ids((size_t)W);
        const bool two = x_assign != x_store;
        for (int64_t i = 0; i < rows; i++) {
            const int64_t l = assign[(size_t)i];
            if (l < 0 || l >= nlist_) return KNHIP_ERR_INVALID_ARGS;  // (a NaN row: IndexIVF::add_core skips it; refused here)
            const int r = owner_[(size_t)l];
            xs[r].insert(xs[r].end(), x_store + i * dim_, x_store + (i + 1) * dim_);
            if (two) xa[r].insert(xa[r].end(), x_assign + i * dim_, x_assign + (i + 1) * dim_);
            ids[r].push_back(n0 + i);
        }
        std::vector<int> rcs((size_t)W, KNHIP_OK);
        std::vector<std::string> errs((size_t)W);
        std::vector<std::thread> th;
        for (int r = 0; r < W; r++) {
            th.emplace_back([&, r]() {
                const int64_t n = (int64_t)ids[r].size();
                if (n == 0) {
                    // A shard that has received nothing yet still needs an (empty) list layout: without one it answers
                    // Search() with "empty index" and fails the whole group (fewer non-empty lists than devices: small
                    // segments, nlist shrunk to rows / 39, a skewed first batch).  Same call Deserialize makes.
                    if (knhip_index_count(sh_[r].idx.p) == 0) {
                        const std::vector<int64_t> zero((size_t)nlist_, 0);
                        const std::vector<const uint8_t*> cp((size_t)nlist_, nullptr);
                        const std::vector<const int64_t*> ip((size_t)nlist_, nullptr);
                        rcs[r] = knhip_index_add_lists(sh_[r].idx.p, zero.data(), cp.data(), ip.data());
                        if (rcs[r]) errs[r] = knhip_last_error();
                    }
                    return;
                }
                rcs[r] = two ? knhip_index_add_assigned_by(sh_[r].idx.p, n, xs[r].data(), xa[r].data(), ids[r].data())
                             : knhip_index_add(sh_[r].idx.p, n, xs[r].data(), ids[r].data());
                if (rcs[r]) errs[r] = knhip_last_error();  // (thread-local text: fetched on the thread that failed)
            });
        }
        for (auto& t : th) t.join();
        for (int r = 0; r < W; r++) {
            if (rcs[r]) {
                LOG_KNOWHERE_ERROR_ << TypeName() << ": add on device " << sh_[r].device << ": " << errs[r];
                return rcs[r];
            }
        }
        return KNHIP_OK;
    }

    // row_scale_by_id_ (ids are the running row numbers) -> every shard's canonical entry order -> knhip_index_set_row_scale
    Status
    PushRowScale() {
        for (auto& s : sh_) {
            const int64_t count = knhip_index_count(s.idx.p);
            if (count <= 0) continue;
            std::vector<float> canon((size_t)count);
            if constexpr (Kind == KNHIP_BRUTE_FORCE) {
                if (s.row_base < 0 || s.row_base + count > (int64_t)row_scale_by_id_.size()) return Status::invalid_args;
                std::copy_n(row_scale_by_id_.begin() + s.row_base, count, canon.begin());
            } else {
                std::vector<int64_t> ids((size_t)count);
                if (int rc = knhip_index_get_lists(s.idx.p, nullptr, ids.data())) return ToStatus(rc);
                for (int64_t i = 0; i < count; i++) {
                    if (ids[i] < 0 || ids[i] >= (int64_t)row_scale_by_id_.size()) return Status::invalid_args;
                    canon[i] = row_scale_by_id_[(size_t)ids[i]];
                }
            }
            if (int rc = knhip_index_set_row_scale(s.idx.p, canon.data(), Kind == KNHIP_BRUTE_FORCE ? 2 : 1)) return ToStatus(rc);
        }
        return Status::success;
    }

    int metric_ = KNHIP_L2;
    bool cosine_ = false, has_refine_ = false;
    int32_t refine_rows_type_ = 0;  // 0: fp32 rows (IndexRefineFlat); KNHIP_ROWS_*: a quantised store (single device)
    std::vector<float> row_scale_by_id_;  // StoredNormCosine(): FLAT inverse L2 norms, IVF_FLAT L2 norms, by row id
    std::string metric_name_ = metric::L2;
    int64_t dim_ = 0, nlist_ = 0, m_ = 0, default_nprobe_ = 8;
    int64_t nbits_ = 8;  // IVF_PQ: code width (1 .. 8)
    std::vector<Shard> sh_;        // one entry: the whole index on one device; several: list- (FLAT: row-) sharded
    GroupHandle group_;            // several shards: the exchange + merge host (include/knhip_shards.h)
    std::vector<int32_t> owner_;   // several shards, IVF kinds: list -> shard
    mutable std::shared_mutex rw_;  // Add / Deserialize (exclusive) vs Search / Serialize (shared)
};

template <typename DataType>
using HipBruteForceIndexNode = HipIndexNode<DataType, KNHIP_BRUTE_FORCE>;
template <typename DataType>
using HipIvfFlatIndexNode = HipIndexNode<DataType, KNHIP_IVF_FLAT>;
template <typename DataType>
using HipIvfPqIndexNode = HipIndexNode<DataType, KNHIP_IVF_PQ>;
template <typename DataType>
using HipIvfSqIndexNode = HipIndexNode<DataType, KNHIP_IVF_SQ8>;

// static-init registration, as src/index/gpu_cuvs/gpu_cuvs_ivf_pq.cc:27-63 does (fp16 / bf16 / int8: below)
KNOWHERE_REGISTER_GLOBAL_WITH_THREAD_POOL(GPU_HIP_BRUTE_FORCE, HipBruteForceIndexNode, fp32,
                                          knowhere::feature::GPU_KNN_FLOAT_INDEX, HipSearchPoolSize());
KNOWHERE_REGISTER_GLOBAL_WITH_THREAD_POOL(GPU_HIP_IVF_FLAT, HipIvfFlatIndexNode, fp32,
                                          knowhere::feature::GPU_ANN_FLOAT_INDEX, HipSearchPoolSize());
KNOWHERE_REGISTER_GLOBAL_WITH_THREAD_POOL(GPU_HIP_IVF_PQ, HipIvfPqIndexNode, fp32,
                                          knowhere::feature::GPU_ANN_FLOAT_INDEX, HipSearchPoolSize());
KNOWHERE_REGISTER_GLOBAL_WITH_THREAD_POOL(GPU_HIP_IVF_SQ8, HipIvfSqIndexNode, fp32,
                                          knowhere::feature::GPU_ANN_FLOAT_INDEX, HipSearchPoolSize());


#if defined(KNHIP_WITH_KNOWHERE_HEADERS)
// fp16 / bf16 / int8 vectors, as the CPU IVF nodes take them (KNOWHERE_MOCK_REGISTER_DENSE_FLOAT_ALL_GLOBAL /
// _DENSE_INT_GLOBAL, ivf.cc:1924-1966): the reference's own IndexNodeDataMockWrapper (index_node_data_mock_wrapper.h: the
// dataset is converted to fp32 on the host, results converted back where vectors are returned) in front of the fp32
// node, inside the GPU nodes' thread-pool wrapper -- KNOWHERE_MOCK_REGISTER_GLOBAL (index_factory.h:95-103) with the pool
// of KNOWHERE_REGISTER_GLOBAL_WITH_THREAD_POOL (:157-165).  Only in a Knowhere tree: the wrapper is the reference's
// code (src/index/index_node_data_mock_wrapper.cc), not restated in the stand-alone shim build.
#define KNHIP_MOCK_REGISTER_WITH_THREAD_POOL(name, index_node, data_type, features, thread_size)                          \
    KNOWHERE_REGISTER_STATIC(name, index_node, data_type)                                                                  \
    KNOWHERE_REGISTER_GLOBAL(                                                                                              \
        name,                                                                                                              \
        [](const int32_t& version, const Object& object) {                                                                 \
            return (Index<IndexNodeThreadPoolWrapper>::Create(                                                             \
                std::make_unique<IndexNodeDataMockWrapper<data_type>>(                                                     \
                    std::make_unique<index_node<MockData<data_type>::type>>(version, object)),                             \
                thread_size));                                                                                             \
        },                                                                                                                 \
        data_type, typeCheck<data_type>(features), features)
#define KNHIP_MOCK_REGISTER_TYPES(name, index_node)                                                                         \
    KNHIP_MOCK_REGISTER_WITH_THREAD_POOL(name, index_node, fp16, knowhere::feature::GPU | knowhere::feature::FP16,         \
                                         HipSearchPoolSize());                                                             \
    KNHIP_MOCK_REGISTER_WITH_THREAD_POOL(name, index_node, bf16, knowhere::feature::GPU | knowhere::feature::BF16,         \
                                         HipSearchPoolSize());                                                             \
    KNHIP_MOCK_REGISTER_WITH_THREAD_POOL(name, index_node, int8, knowhere::feature::GPU | knowhere::feature::INT8,         \
                                         HipSearchPoolSize());
KNHIP_MOCK_REGISTER_TYPES(GPU_HIP_BRUTE_FORCE, HipBruteForceIndexNode)
KNHIP_MOCK_REGISTER_TYPES(GPU_HIP_IVF_FLAT, HipIvfFlatIndexNode)
KNHIP_MOCK_REGISTER_TYPES(GPU_HIP_IVF_PQ, HipIvfPqIndexNode)
KNHIP_MOCK_REGISTER_TYPES(GPU_HIP_IVF_SQ8, HipIvfSqIndexNode)
#endif

}  // namespace knowhere

// test hook (node_capi.cc): where the last index built / loaded on this thread lives
extern "C" int32_t
knhip_host_last_placement(int32_t* out, int32_t cap) {
    const auto& d = knowhere::g_last_placement;
    for (int32_t i = 0; i < (int32_t)d.size() && i < cap; i++) out[i] = d[(size_t)i];
    return (int32_t)d.size();
}

// test hook (node_capi.cc): the node's NormalizeVec restatement
extern "C" void
knhip_host_normalize_rows(float* x, int64_t n, int64_t d, float* norms) {
    for (int64_t i = 0; i < n; i++) {
        const float nr = knowhere::NormalizeRow(x + i * d, d);
        if (norms) norms[i] = nr;
    }
}
