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
//   Serialize / Deserialize: one named blob (Type()) in a BinarySet (ivf.cc:1717-1834) in the FAISS byte format the
//           CPU nodes write (IxF2/IxFI/IxF9, IwFl, IwSq, IwPQ, IxRF; faiss_io.h), so a CPU-built index loads here and back.
#include "hip_index_node.h"

#include "../../include/knhip.h"
#include "faiss_io.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <numeric>
#include <shared_mutex>

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
    ~KnhipHandle() { knhip_index_destroy(p); }
    void reset() {
        knhip_index_destroy(p);
        p = nullptr;
    }
};

}  // namespace

template <typename DataType, int Kind>
class HipIndexNode : public IndexNode {
    static_assert(std::is_same_v<DataType, fp32>, "fp16 / bf16 / int8 come through the reference's mock wrapper "
                                                  "(KNOWHERE_MOCK_REGISTER_GLOBAL, index_factory.h:95-103)");

 public:
    using knowhere_config_type =
        std::conditional_t<Kind == KNHIP_BRUTE_FORCE, HipBruteForceConfig,
                           std::conditional_t<Kind == KNHIP_IVF_FLAT, HipIvfFlatConfig,
                                              std::conditional_t<Kind == KNHIP_IVF_PQ, HipIvfPqConfig, HipIvfSqConfig>>>;

    HipIndexNode(const int32_t& /*version*/, const Object& /*object*/) {}
    ~HipIndexNode() override = default;

    bool
    NeedBitsetExactCount() const override {
        return true;  // the all-filtered shortcut needs the exact projected count (gpu_cuvs.h:82-85)
    }

    Status
    Train(const DataSetPtr dataset, std::shared_ptr<Config> cfg, bool /*use_knowhere_build_pool*/) override {
        if (!dataset || !dataset->GetTensor() || !cfg) return Status::invalid_args;
        if (idx_.p) return Status::index_already_trained;
        const auto& c = static_cast<const knowhere_config_type&>(*cfg);
        const int64_t rows = dataset->GetRows();
        dim_ = dataset->GetDim();
        if (c.dim.has_value() && c.dim.value() != dim_) return Status::invalid_args;
        const std::string metric = c.metric_type.value_or(metric::L2);
        cosine_ = IsMetricType(metric, metric::COSINE);
        metric_ = IsMetricType(metric, metric::L2) ? KNHIP_L2 : KNHIP_IP;
        metric_name_ = cosine_ ? metric::COSINE : (metric_ == KNHIP_L2 ? metric::L2 : metric::IP);
        knhip_desc desc{};
        desc.kind = Kind;
        desc.metric = metric_;
        desc.dim = (int32_t)dim_;
        if constexpr (Kind != KNHIP_BRUTE_FORCE) {
            // MatchNlist: silently shrink nlist so that nlist * 39 <= rows (ivf.cc:478-489)
            nlist_ = c.nlist.value();
            if (nlist_ * 39 > rows) nlist_ = std::max<int64_t>(1, rows / 39);
            desc.nlist = nlist_;
            default_nprobe_ = c.nprobe.value_or(8);
        }
        if constexpr (Kind == KNHIP_IVF_PQ) {
            // m = 0: the backend picks, as cuVS does for pq_dim = 0 (about dim / 2): the largest supported m that
            // leaves sub-vectors of at least 2 dims
            // leaves sub-vectors of at least 2 dims.  An explicit m is honoured or refused, never rewritten (the reference
            // honours any m that divides dim; this backend has kernels for 8, 16, 32 and 64 sub-quantizers)
            const bool auto_m = c.m.value_or(0) == 0;
            m_ = auto_m ? std::min<int64_t>(64, dim_ / 2) : c.m.value();
            while (auto_m && m_ > 1 && (dim_ % m_ != 0 || !(m_ == 8 || m_ == 16 || m_ == 32 || m_ == 64))) m_--;
            if (dim_ % m_ != 0 || !(m_ == 8 || m_ == 16 || m_ == 32 || m_ == 64)) return Status::invalid_args;
            desc.pq_m = (int32_t)m_;
            desc.pq_nbits = 8;
        }
        if constexpr (Kind == KNHIP_IVF_PQ || Kind == KNHIP_IVF_SQ8) {
            has_refine_ = c.refine.value_or(false);
        }
        int rc = knhip_index_create(&desc, &idx_.p);
        if (rc) return ToStatus(rc);
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
        rc = knhip_index_train(idx_.p, rows, x, nullptr);  // the reference's clustering defaults
        if (rc) {
            idx_.reset();
            return ToStatus(rc);
        }
        return Status::success;
    }

    // thread safe against concurrent Search (index_node.h:141-145): the index object serialises layout changes
    Status
    Add(const DataSetPtr dataset, std::shared_ptr<Config> /*cfg*/, bool /*use_knowhere_build_pool*/) override {
        if (!dataset || !dataset->GetTensor()) return Status::invalid_args;
        if (!idx_.p) return Kind == KNHIP_BRUTE_FORCE ? Status::empty_index : Status::index_not_trained;
        if (dataset->GetDim() != dim_) return Status::invalid_args;
        const int64_t rows = dataset->GetRows();
        const float* x = (const float*)dataset->GetTensor();
        std::vector<float> xn;
        if (StoredNormCosine()) {
            // FLAT / IVF_FLAT keep the RAW rows and one float per row beside them, as IndexFlatCosine (inverse norms,
            // IndexCosine.cpp:236-245) and IndexIVFFlatCosine (norms, IndexIVFFlat.cpp:516-524: assigned by the
            // normalised row) do: the scores are then the CPU node's floats
            std::unique_lock<std::shared_mutex> lk(rw_);
            const size_t n0 = row_scale_by_id_.size();
            row_scale_by_id_.resize(n0 + (size_t)rows);
            int rc = 0;
            if constexpr (Kind == KNHIP_BRUTE_FORCE) {
                for (int64_t i = 0; i < rows; i++) row_scale_by_id_[n0 + i] = InverseL2Norm(x + i * dim_, dim_);
                rc = knhip_index_add(idx_.p, rows, x, nullptr);
            } else {
                xn.assign(x, x + rows * dim_);
                NormalizeRows(xn.data(), rows, dim_, row_scale_by_id_.data() + n0);
                rc = knhip_index_add_assigned_by(idx_.p, rows, x, xn.data(), nullptr);
            }
            if (rc) {
                row_scale_by_id_.resize(n0);
                return ToStatus(rc);
            }
            return PushRowScale();
        }
        if (cosine_) {
            xn.assign(x, x + rows * dim_);
            NormalizeRows(xn.data(), rows, dim_);
            x = xn.data();
        }
        std::unique_lock<std::shared_mutex> lk(rw_);
        int rc = knhip_index_add(idx_.p, rows, x, nullptr);
        if (rc) return ToStatus(rc);
        if (NeedRawStore()) {
            if (!raw_.p) {
                knhip_desc rd{};
                rd.kind = KNHIP_BRUTE_FORCE;
                rd.metric = metric_;
                rd.dim = (int32_t)dim_;
                if ((rc = knhip_index_create(&rd, &raw_.p))) return ToStatus(rc);
            }
            if ((rc = knhip_index_add(raw_.p, rows, x, nullptr))) return ToStatus(rc);
        }
        return Status::success;
    }

    expected<DataSetPtr>
    Search(const DataSetPtr dataset, std::unique_ptr<Config> cfg, const BitsetView& bitset,
           milvus::OpContext* op_context) const override {
        if (!idx_.p || Count() == 0) return expected<DataSetPtr>::Err(Status::empty_index, "index not built");
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
        const bool has_bitset = bitset.data() != nullptr && bitset.num_bits() != 0;
        if (has_bitset && bitset.has_out_ids()) {
            const size_t n_in = bitset.out_ids_count();
            in_bitset.assign((n_in + 7) / 8, 0);
            for (size_t i = 0; i < n_in; i++) {
                if (bitset.test((int64_t)i)) in_bitset[i >> 3] |= (uint8_t)(1u << (i & 7));
            }
            bits = in_bitset.data();
            nbits = (int64_t)n_in;
        } else if (has_bitset) {
            bits = bitset.data();
            nbits = (int64_t)bitset.num_bits();
        }
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
        {
            std::shared_lock<std::shared_mutex> lk(rw_);
            // use_refine = the index carries a refine index (ivf.cc:1080-1092); k_factor = refine_k
            int64_t kbase = k;
            if constexpr (Kind == KNHIP_IVF_PQ || Kind == KNHIP_IVF_SQ8) {
                if (has_refine_ && raw_.p && c.refine_k.has_value()) {
                    const int64_t want = std::max<int64_t>(k, (int64_t)(k * c.refine_k.value()));
                    kbase = std::min<int64_t>(1024, want);
                    if (kbase < want) {  // (the first stage returns at most 1024 candidates per query: said, not hidden)
                        LOG_KNOWHERE_WARNING_ << "GPU_HIP refine: k * refine_k = " << want << " clamped to " << kbase
                                              << (kbase <= k ? " (refine stage skipped)" : "");
                    }
                }
            }
            if (kbase > k) {
                rc = knhip_search_refine(idx_.p, raw_.p, q, nq, (int32_t)k, (int32_t)kbase, (int32_t)nprobe, bits, nbits,
                                         ids.get(), dis.get());
            } else {
                rc = knhip_search(idx_.p, q, nq, (int32_t)k, (int32_t)nprobe, bits, nbits, ids.get(), dis.get());
            }
        }
        if (rc) return expected<DataSetPtr>::Err(ToStatus(rc), knhip_last_error());
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
        if (!idx_.p || Count() == 0) return expected<DataSetPtr>::Err(Status::empty_index, "index not built");
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
        const bool has_bitset = bitset.data() != nullptr && bitset.num_bits() != 0;
        if (has_bitset && bitset.has_out_ids()) {
            const size_t n_in = bitset.out_ids_count();
            in_bitset.assign((n_in + 7) / 8, 0);
            for (size_t i = 0; i < n_in; i++) {
                if (bitset.test((int64_t)i)) in_bitset[i >> 3] |= (uint8_t)(1u << (i & 7));
            }
            bits = in_bitset.data();
            nbits = (int64_t)n_in;
        } else if (has_bitset) {
            bits = bitset.data();
            nbits = (int64_t)bitset.num_bits();
        }
        std::vector<int64_t> lims(nq + 1);
        int64_t* ids = nullptr;
        float* dis = nullptr;
        int rc;
        {
            std::shared_lock<std::shared_mutex> lk(rw_);
            rc = knhip_range_search(idx_.p, q, nq, radius, (int32_t)max_empty, bits, nbits, lims.data(), &ids, &dis);
        }
        if (rc) return expected<DataSetPtr>::Err(ToStatus(rc), knhip_last_error());
        const bool is_ip = metric_ != KNHIP_L2;
        auto out_lims = std::make_unique<size_t[]>(nq + 1);
        auto out_ids = std::make_unique<int64_t[]>(std::max<int64_t>(lims[nq], 1));
        auto out_dis = std::make_unique<float[]>(std::max<int64_t>(lims[nq], 1));
        size_t n = 0;
        out_lims[0] = 0;
        for (int64_t i = 0; i < nq; i++) {
            for (int64_t j = lims[i]; j < lims[i + 1]; j++) {
                const float v = dis[j];
                const bool keep = range_filter == defaultRangeFilter ||
                                  (is_ip ? (radius < v && v <= range_filter) : (range_filter <= v && v < radius));
                if (keep) {
                    out_ids[n] = ids[j];
                    out_dis[n] = v;
                    n++;
                }
            }
            out_lims[i + 1] = n;
        }
        knhip_free(ids);
        knhip_free(dis);
        auto res = GenResultDataSet(nq, out_ids.release(), out_dis.release(), out_lims.release());
        this->MapSearchResultIdsToOutIds(res);
        return res;
    }

    expected<DataSetPtr>
    GetVectorByIds(const DataSetPtr dataset, milvus::OpContext* /*op_context*/) const override {
        if (!dataset || !dataset->GetIds()) return expected<DataSetPtr>::Err(Status::invalid_args, "null ids");
        if (!HasRawData(metric_name_)) return expected<DataSetPtr>::Err(Status::not_implemented, "no raw data");
        // (IVF_FLAT: the index's own rows through its direct map -- no second copy of the raw vectors)
        const knhip_index* store = idx_.p;
        if (!store) return expected<DataSetPtr>::Err(Status::empty_index, "index not built");
        const int64_t n = dataset->GetRows();
        auto out = std::make_unique<float[]>(std::max<int64_t>(n * dim_, 1));
        std::shared_lock<std::shared_mutex> lk(rw_);
        const int rc = knhip_index_get_vectors(store, n, dataset->GetIds(), out.get());
        if (rc) return expected<DataSetPtr>::Err(ToStatus(rc), knhip_last_error());
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
    // state and the inverted lists are read back from the device.
    Status
    Serialize(BinarySet& binset) const override {
        if (!idx_.p) return Status::empty_index;
        using namespace knhip_host;
        std::shared_lock<std::shared_mutex> lk(rw_);
        const int64_t count = knhip_index_count(idx_.p);
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
        if constexpr (Kind == KNHIP_BRUTE_FORCE) {
            x.xb.resize((size_t)count * dim_);
            if ((rc = knhip_index_get_lists(idx_.p, (uint8_t*)x.xb.data(), nullptr))) return ToStatus(rc);
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
            if ((rc = knhip_index_get_coarse(idx_.p, x.quantizer.xb.data()))) return ToStatus(rc);
            x.by_residual = true;
            x.code_size = (uint64_t)CodeSize();
            if constexpr (Kind == KNHIP_IVF_PQ) {
                x.pq_d = (uint64_t)dim_;
                x.pq_M = (uint64_t)m_;
                x.pq_nbits = 8;
                x.pq_centroids.resize((size_t)256 * dim_);
                if ((rc = knhip_index_get_pq(idx_.p, x.pq_centroids.data()))) return ToStatus(rc);
            } else if constexpr (Kind == KNHIP_IVF_SQ8) {
                x.sq_qtype = 0;      // ScalarQuantizer::QT_8bit
                x.sq_rangestat = 0;  // RS_minmax
                x.sq_d = (uint64_t)dim_;
                x.sq_code_size = (uint64_t)dim_;
                x.sq_trained.resize((size_t)2 * dim_);
                if ((rc = knhip_index_get_sq(idx_.p, x.sq_trained.data(), x.sq_trained.data() + dim_))) return ToStatus(rc);
            }
            std::vector<int64_t> sizes((size_t)nlist_);
            if ((rc = knhip_index_get_list_sizes(idx_.p, sizes.data()))) return ToStatus(rc);
            std::vector<uint8_t> codes((size_t)count * CodeSize());
            std::vector<int64_t> ids((size_t)count);
            if ((rc = knhip_index_get_lists(idx_.p, codes.data(), ids.data()))) return ToStatus(rc);
            x.codes.assign(nlist_, {});
            x.ids.assign(nlist_, {});
            size_t non0 = 0;
            int64_t pos = 0;
            for (int64_t l = 0; l < nlist_; l++) {
                x.codes[l].assign(codes.begin() + pos * CodeSize(), codes.begin() + (pos + sizes[l]) * CodeSize());
                x.ids[l].assign(ids.begin() + pos, ids.begin() + pos + sizes[l]);
                pos += sizes[l];
                non0 += sizes[l] != 0;
            }
            x.lists_sparse = !(non0 > (size_t)nlist_ / 2);  // index_write.cpp:309-316
            if (Kind == KNHIP_IVF_FLAT && cosine_) {        // Knowhere cosine IVF-Flat carries the row norms
                x.with_norm = true;
                x.norms.assign(nlist_, {});
                for (int64_t l = 0; l < nlist_; l++) {
                    x.norms[l].resize((size_t)sizes[l]);
                    for (int64_t j = 0; j < sizes[l]; j++) x.norms[l][(size_t)j] = row_scale_by_id_[(size_t)x.ids[l][(size_t)j]];
                }
            }
            if (has_refine_ && raw_.p) {  // IndexRefineFlat (ivf.cc:673-700)
                x.has_refine = true;
                fill_hdr(x.refine_hdr, count, cosine_);
                x.refine_index.fourcc = flat_cc;
                fill_hdr(x.refine_index.hdr, count, false);
                x.refine_index.xb.resize((size_t)count * dim_);
                if ((rc = knhip_index_get_lists(raw_.p, (uint8_t*)x.refine_index.xb.data(), nullptr))) return ToStatus(rc);
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
            const uint64_t want_cs = Kind == KNHIP_IVF_FLAT ? (uint64_t)d * 4 : Kind == KNHIP_IVF_PQ ? x.pq_M : (uint64_t)d;
            if (Kind == KNHIP_IVF_PQ &&
                (x.pq_nbits != 8 || !x.by_residual || !(x.pq_M == 8 || x.pq_M == 16 || x.pq_M == 32 || x.pq_M == 64)))
                return Status::not_implemented;
            if (Kind == KNHIP_IVF_PQ && (x.pq_d != (uint64_t)d || d % (int64_t)x.pq_M != 0 ||
                                          x.pq_centroids.size() != (size_t)256 * d))
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
                if (x.refine_index.hdr.ntotal != ntotal || x.refine_index.hdr.d != d ||
                    x.refine_index.xb.size() != (size_t)ntotal * d)
                    return Status::invalid_serialized_index_type;
                for (uint64_t l = 0; l < x.nlist; l++) {
                    for (int64_t id : x.ids[l]) {
                        if (id < 0 || id >= ntotal) return Status::invalid_serialized_index_type;
                    }
                }
            }
        }
        metric_ = x.hdr.metric == 1 ? KNHIP_L2 : KNHIP_IP;
        cosine_ = x.hdr.is_cosine() || x.fourcc == FourCC("IxF9") || x.with_norm;
        if (cfg) {
            const auto& c = static_cast<const BaseConfig&>(*cfg);
            if (c.metric_type.has_value() && IsMetricType(c.metric_type.value(), metric::COSINE) && metric_ == KNHIP_IP)
                cosine_ = true;
        }
        metric_name_ = cosine_ ? metric::COSINE : (metric_ == KNHIP_L2 ? metric::L2 : metric::IP);
        dim_ = d;
        nlist_ = (int64_t)x.nlist;
        if (x.nprobe >= 1 && x.nprobe <= 65536) default_nprobe_ = (int64_t)x.nprobe;  // the index's default nprobe
        m_ = (int64_t)x.pq_M;
        has_refine_ = x.has_refine;
        // The CPU cosine indexes keep the RAW rows plus their L2 norms (FLAT: the wire carries the norms, the index
        // multiplies by their inverses, L2NormsStorage::add_l2_norms IndexCosine.cpp:247-255; IVF_FLAT: ip / norm,
        // cppcontrib/knowhere/IndexIVFFlat.cpp:199-210): kept exactly so, per row id
        std::vector<float> scale_by_id;
        if (cosine_ && flat) {
            if (x.flat_norms.size() != (size_t)ntotal) return Status::invalid_serialized_index_type;
            scale_by_id.resize((size_t)ntotal);
            for (int64_t i = 0; i < ntotal; i++) {
                const float nr = x.flat_norms[(size_t)i];
                scale_by_id[(size_t)i] = nr == 0.0f ? 1.0f : (1.0f / nr);
            }
        }
        if (cosine_ && Kind == KNHIP_IVF_FLAT) {
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
        idx_.reset();
        raw_.reset();
        row_scale_by_id_ = std::move(scale_by_id);
        knhip_desc desc{};
        desc.kind = Kind;
        desc.metric = metric_;
        desc.dim = (int32_t)dim_;
        desc.nlist = nlist_;
        desc.pq_m = (int32_t)m_;
        desc.pq_nbits = 8;
        int rc = knhip_index_create(&desc, &idx_.p);
        if (rc) return ToStatus(rc);
        if constexpr (Kind == KNHIP_BRUTE_FORCE) {
            if ((rc = knhip_index_add(idx_.p, ntotal, x.xb.data(), nullptr))) return ToStatus(rc);
            return StoredNormCosine() ? PushRowScale() : Status::success;
        }
        if ((rc = knhip_index_set_coarse(idx_.p, x.quantizer.xb.data()))) return ToStatus(rc);
        if (Kind == KNHIP_IVF_PQ && (rc = knhip_index_set_pq(idx_.p, x.pq_centroids.data()))) return ToStatus(rc);
        if (Kind == KNHIP_IVF_SQ8 && (rc = knhip_index_set_sq(idx_.p, x.sq_trained.data(), x.sq_trained.data() + dim_)))
            return ToStatus(rc);
        std::vector<int64_t> sizes(nlist_);
        std::vector<const uint8_t*> cp(nlist_);
        std::vector<const int64_t*> ip(nlist_);
        for (int64_t l = 0; l < nlist_; l++) {
            sizes[l] = (int64_t)x.ids[l].size();
            cp[l] = x.codes[l].data();
            ip[l] = x.ids[l].data();
        }
        if ((rc = knhip_index_add_lists(idx_.p, sizes.data(), cp.data(), ip.data()))) return ToStatus(rc);
        if (StoredNormCosine()) {
            const Status st = PushRowScale();
            if (st != Status::success) return st;
        }
        // raw rows for refine, in id order
        if (x.has_refine && !x.refine_index.xb.empty()) {
            knhip_desc rd{};
            rd.kind = KNHIP_BRUTE_FORCE;
            rd.metric = metric_;
            rd.dim = (int32_t)dim_;
            if ((rc = knhip_index_create(&rd, &raw_.p))) return ToStatus(rc);
            if ((rc = knhip_index_add(raw_.p, ntotal, x.refine_index.xb.data(), nullptr))) return ToStatus(rc);
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
        // what the typed config cannot express: this backend needs at least one visible device
        if (paramType == PARAM_TYPE::TRAIN && knhip_device_count() <= 0) {
            msg = "no HIP device available for " + std::string(TypeName());
            return Status::cuda_runtime_error;
        }
        return Status::success;
    }

    int64_t
    Dim() const override {
        return dim_;
    }
    int64_t
    Size() const override {
        return (idx_.p ? knhip_index_device_bytes(idx_.p) : 0) + (raw_.p ? knhip_index_device_bytes(raw_.p) : 0);
    }
    int64_t
    Count() const override {
        return idx_.p ? knhip_index_count(idx_.p) : 0;
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

 private:
    int64_t
    CodeSize() const {
        return Kind == KNHIP_IVF_FLAT ? dim_ * 4 : (Kind == KNHIP_IVF_PQ ? m_ : dim_);
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
    // row_scale_by_id_ (ids are the running row numbers) -> the index's canonical entry order -> knhip_index_set_row_scale
    Status
    PushRowScale() {
        const int64_t count = knhip_index_count(idx_.p);
        if (count <= 0) return Status::success;
        std::vector<float> canon((size_t)count);
        if constexpr (Kind == KNHIP_BRUTE_FORCE) {
            if ((int64_t)row_scale_by_id_.size() != count) return Status::invalid_args;
            canon = row_scale_by_id_;
        } else {
            std::vector<int64_t> ids((size_t)count);
            if (int rc = knhip_index_get_lists(idx_.p, nullptr, ids.data())) return ToStatus(rc);
            for (int64_t i = 0; i < count; i++) {
                if (ids[i] < 0 || ids[i] >= (int64_t)row_scale_by_id_.size()) return Status::invalid_args;
                canon[i] = row_scale_by_id_[(size_t)ids[i]];
            }
        }
        return ToStatus(knhip_index_set_row_scale(idx_.p, canon.data(), Kind == KNHIP_BRUTE_FORCE ? 2 : 1));
    }

    int metric_ = KNHIP_L2;
    bool cosine_ = false, has_refine_ = false;
    std::vector<float> row_scale_by_id_;  // StoredNormCosine(): FLAT inverse L2 norms, IVF_FLAT L2 norms, by row id
    std::string metric_name_ = metric::L2;
    int64_t dim_ = 0, nlist_ = 0, m_ = 0, default_nprobe_ = 8;
    KnhipHandle idx_, raw_;
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

// static-init registration, as src/index/gpu_cuvs/gpu_cuvs_ivf_pq.cc:27-63 does (fp16 / bf16 / int8: add the matching
// KNOWHERE_MOCK_REGISTER_GLOBAL lines)
KNOWHERE_REGISTER_GLOBAL_WITH_THREAD_POOL(GPU_HIP_BRUTE_FORCE, HipBruteForceIndexNode, fp32,
                                          knowhere::feature::GPU_KNN_FLOAT_INDEX, HipSearchPoolSize());
KNOWHERE_REGISTER_GLOBAL_WITH_THREAD_POOL(GPU_HIP_IVF_FLAT, HipIvfFlatIndexNode, fp32,
                                          knowhere::feature::GPU_ANN_FLOAT_INDEX, HipSearchPoolSize());
KNOWHERE_REGISTER_GLOBAL_WITH_THREAD_POOL(GPU_HIP_IVF_PQ, HipIvfPqIndexNode, fp32,
                                          knowhere::feature::GPU_ANN_FLOAT_INDEX, HipSearchPoolSize());
KNOWHERE_REGISTER_GLOBAL_WITH_THREAD_POOL(GPU_HIP_IVF_SQ8, HipIvfSqIndexNode, fp32,
                                          knowhere::feature::GPU_ANN_FLOAT_INDEX, HipSearchPoolSize());

}  // namespace knowhere

// test hook (node_capi.cc): the node's NormalizeVec restatement
extern "C" void
knhip_host_normalize_rows(float* x, int64_t n, int64_t d, float* norms) {
    for (int64_t i = 0; i < n; i++) {
        const float nr = knowhere::NormalizeRow(x + i * d, d);
        if (norms) norms[i] = nr;
    }
}
