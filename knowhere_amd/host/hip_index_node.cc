// knowhere_amd/host/hip_index_node.cc -- the Knowhere IndexNode of the MI355X backend.
//
// Modelled on GpuCuvsIndexNode (reference src/index/gpu_cuvs/gpu_cuvs.h:73-324) and registered
// the same way (src/index/gpu_cuvs/gpu_cuvs_ivf_pq.cc:27-63) under NEW index names:
//     GPU_HIP_BRUTE_FORCE, GPU_HIP_IVF_FLAT, GPU_HIP_IVF_PQ, GPU_HIP_IVF_SQ8
// It owns no arithmetic: everything numeric goes through the C ABI of libknhip.so
// (include/knhip.h).  Mapping to the reference:
//   Train   IvfIndexNode::Train (src/index/ivf/ivf.cc:547-807): MatchNlist (>= 39 points per
//           centroid, :478-489), k-means for the coarse quantizer, PQ codebooks on residuals
//           (faiss IndexIVFPQ::train_encoder), SQ8 min/max ranges (RS_minmax).  Lloyd iterations run
//           on the host; the expensive step -- nearest-centroid assignment -- is a GPU brute-force
//           search (k = 1) through the same C ABI.
//   Add     IvfIndexNode::Add (ivf.cc:811-844) -> IndexIVF::add_core: assign, encode the residual,
//           append (code, id) to the list; ids are the running row numbers.
//   Search  GpuCuvsIndexNode::Search (gpu_cuvs.h:121-190): config -> knhip_search -> GenResultDataSet
//           (takes ownership of two new[] arrays) ; `refine` / `refine_k` as IvfIndexNode::Search
//           does with IndexRefine (ivf.cc:1073-1103).
//   RangeSearch  IvfIndexNode::RangeSearch (ivf.cc:1231-1497) -> knhip_range_search (brute force, IVF_FLAT,
//           IVF_SQ8, IVF_PQ m = 32; other m and nlist > 4096 report not_implemented).  GetIndexMeta: not_implemented,
//           as the cuVS node (gpu_cuvs.h:192-201).
//   COSINE  base normalised at Train/Add, query copied + normalised per Search, metric -> IP
//           (ivf.cc:559-565, 1068-1071).
//   Serialize / Deserialize: one named blob (Type()) in a BinarySet (ivf.cc:1717-1834) in the FAISS
//           byte format the CPU nodes write (IxF2/IxFI, IwFl, IwSq, IwPQ, IxRF around the latter two
//           when built with `refine`; faiss_io.h), so a CPU-built index loads here unchanged and back.
//   refine  build-time `refine` keeps the fp32 rows (IndexRefineFlat, ivf.cc:673-700); search-time
//           `refine_k` re-ranks only if they are there (ivf.cc:1073-1103).
#include "knowhere_shim.h"
#include "faiss_io.h"

#include "../../include/knhip.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <numeric>
#include <random>

namespace knowhere {

namespace {

Status ToStatus(int rc) {
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

struct HipConfig {  // IvfPqConfig / GpuCuvsIvfPqConfig fields this path consumes (ivf_config.h:33-135)
    int64_t dim = 0, k = 10, nlist = 128, nprobe = 8, m = 32, nbits = 8, refine_k = 0;
    bool refine = false;
    std::string metric = metric::L2;
};

Status LoadConfig(const Json& j, HipConfig& c, bool for_search, std::string* msg) {
    auto get_int = [&](const char* key, int64_t& dst, int64_t lo, int64_t hi) -> Status {
        if (!j.contains(key)) return Status::success;
        const JsonValue& v = j.at(key);
        if (!v.is_number()) {
            *msg = std::string("type conflict for ") + key;
            return Status::type_conflict_in_json;
        }
        const int64_t x = v.as_int();
        if (x < lo || x > hi) {
            *msg = std::string("out of range: ") + key;
            return Status::out_of_range_in_json;
        }
        dst = x;
        return Status::success;
    };
    Status s;
    if ((s = get_int(meta::DIM, c.dim, 1, 32768)) != Status::success) return s;
    if ((s = get_int(meta::TOPK, c.k, 1, 1024)) != Status::success) return s;  // gpu_cuvs_ivf_pq_config.h:49-53
    if ((s = get_int(indexparam::NLIST, c.nlist, 1, 65536)) != Status::success) return s;
    if ((s = get_int(indexparam::NPROBE, c.nprobe, 1, 65536)) != Status::success) return s;
    int64_t m = c.m;
    if ((s = get_int(indexparam::M, m, 0, 65536)) != Status::success) return s;
    c.m = m;
    if ((s = get_int(indexparam::NBITS, c.nbits, 1, 24)) != Status::success) return s;
    if ((s = get_int(indexparam::REFINE_K, c.refine_k, 0, 1024)) != Status::success) return s;
    if (j.contains(indexparam::REFINE)) {
        if (!j.at(indexparam::REFINE).is_boolean()) return Status::type_conflict_in_json;
        c.refine = j.at(indexparam::REFINE).as_bool();
    }
    if (j.contains(meta::METRIC_TYPE)) {
        if (!j.at(meta::METRIC_TYPE).is_string()) return Status::type_conflict_in_json;
        c.metric = j.at(meta::METRIC_TYPE).as_string();
    }
    if (c.metric != metric::L2 && c.metric != metric::IP && c.metric != metric::COSINE) {
        *msg = "metric type " + c.metric + " not supported";
        return Status::invalid_metric_type;
    }
    (void)for_search;
    return Status::success;
}

void NormalizeRows(float* x, int64_t n, int64_t d) {  // CopyAndNormalizeVecs
    for (int64_t i = 0; i < n; i++) {
        float* v = x + i * d;
        double s = 0;
        for (int64_t t = 0; t < d; t++) s += (double)v[t] * v[t];
        const float inv = s > 0 ? (float)(1.0 / std::sqrt(s)) : 0.f;
        for (int64_t t = 0; t < d; t++) v[t] *= inv;
    }
}

struct KnhipHandle {
    knhip_index* p = nullptr;
    ~KnhipHandle() { knhip_index_destroy(p); }
};

// nearest centroid (L2) of every row through a temporary brute-force knhip index
Status AssignGpu(const float* cen, int64_t ncen, const float* x, int64_t n, int d, std::vector<int64_t>& out) {
    knhip_desc desc{};
    desc.kind = KNHIP_BRUTE_FORCE;
    desc.metric = KNHIP_L2;
    desc.dim = d;
    KnhipHandle h;
    int rc = knhip_index_create(&desc, &h.p);
    if (rc) return ToStatus(rc);
    if ((rc = knhip_index_add_vectors(h.p, ncen, cen, nullptr, 0))) return ToStatus(rc);
    out.resize(n);
    std::vector<float> dist(n);
    rc = knhip_search(h.p, x, n, 1, 1, nullptr, 0, out.data(), dist.data());
    return ToStatus(rc);
}

// Lloyd k-means; assignment on the GPU, update on the host (Clustering.h:24-77 defaults: seed 1234)
Status KMeans(const float* x, int64_t n, int d, int64_t k, int niter, std::vector<float>& cen) {
    std::mt19937_64 rng(1234);
    std::vector<int64_t> perm(n);
    std::iota(perm.begin(), perm.end(), 0);
    std::shuffle(perm.begin(), perm.end(), rng);
    cen.resize((size_t)k * d);
    for (int64_t c = 0; c < k; c++) std::memcpy(&cen[c * d], x + perm[c % n] * d, sizeof(float) * d);
    std::vector<int64_t> a;
    std::vector<double> sum((size_t)k * d);
    std::vector<int64_t> cnt(k);
    for (int it = 0; it < niter; it++) {
        Status s = AssignGpu(cen.data(), k, x, n, d, a);
        if (s != Status::success) return s;
        std::fill(sum.begin(), sum.end(), 0.0);
        std::fill(cnt.begin(), cnt.end(), 0);
        for (int64_t i = 0; i < n; i++) {
            const int64_t c = a[i];
            cnt[c]++;
            for (int t = 0; t < d; t++) sum[c * d + t] += x[i * d + t];
        }
        for (int64_t c = 0; c < k; c++) {
            if (cnt[c] == 0) {  // re-seed an empty cluster from a random point
                std::memcpy(&cen[c * d], x + perm[(c * 7919 + it) % n] * d, sizeof(float) * d);
                continue;
            }
            for (int t = 0; t < d; t++) cen[c * d + t] = (float)(sum[c * d + t] / cnt[c]);
        }
    }
    return Status::success;
}

}  // namespace

class HipIndexNode : public IndexNode {
 public:
    HipIndexNode(int32_t /*version*/, int kind) : kind_(kind) {}
    ~HipIndexNode() override { knhip_index_destroy(idx_); }

    Status Train(const DataSetPtr dataset, const Json& cfg) override {
        if (!dataset || !dataset->GetTensor()) return Status::invalid_args;
        if (idx_) return Status::index_already_trained;
        std::string msg;
        Status s = LoadConfig(cfg, cfg_, false, &msg);
        if (s != Status::success) return s;
        const int64_t rows = dataset->GetRows();
        dim_ = dataset->GetDim();
        if (cfg_.dim != 0 && cfg_.dim != dim_) return Status::invalid_args;
        cosine_ = cfg_.metric == metric::COSINE;
        metric_ = (cfg_.metric == metric::L2) ? KNHIP_L2 : KNHIP_IP;
        if (kind_ == KNHIP_BRUTE_FORCE) {
            return Status::success;  // nothing to train
        }
        // MatchNlist: silently shrink nlist so that nlist * 39 <= rows (ivf.cc:478-489)
        nlist_ = cfg_.nlist;
        if (nlist_ * 39 > rows) nlist_ = std::max<int64_t>(1, rows / 39);
        if (kind_ == KNHIP_IVF_PQ) {
            if (cfg_.nbits != 8) return Status::invalid_args;
            // m = 0: let the backend pick, as cuVS does for pq_dim = 0 (about dim / 2): the largest
            // supported m that leaves sub-vectors of at least 2 dims
            m_ = cfg_.m == 0 ? std::min<int64_t>(64, dim_ / 2) : cfg_.m;
            while (m_ > 1 && (dim_ % m_ != 0 || !(m_ == 8 || m_ == 16 || m_ == 32 || m_ == 64))) m_--;
            if (dim_ % m_ != 0 || !(m_ == 8 || m_ == 16 || m_ == 32 || m_ == 64)) return Status::invalid_args;
        }
        std::vector<float> x((const float*)dataset->GetTensor(), (const float*)dataset->GetTensor() + rows * dim_);
        if (cosine_) NormalizeRows(x.data(), rows, dim_);
        // train on at most 256 points per centroid (Clustering.h:46)
        const int64_t ntrain = std::min<int64_t>(rows, 256 * nlist_);
        s = KMeans(x.data(), ntrain, (int)dim_, nlist_, 10, centroids_);
        if (s != Status::success) return s;
        std::vector<int64_t> a;
        if ((s = AssignGpu(centroids_.data(), nlist_, x.data(), ntrain, (int)dim_, a)) != Status::success) return s;
        std::vector<float> resid((size_t)ntrain * dim_);
        for (int64_t i = 0; i < ntrain; i++)
            for (int64_t t = 0; t < dim_; t++) resid[i * dim_ + t] = x[i * dim_ + t] - centroids_[a[i] * dim_ + t];
        if (kind_ == KNHIP_IVF_PQ) {
            const int64_t dsub = dim_ / m_;
            codebooks_.assign((size_t)256 * dim_, 0.f);
            std::vector<float> sub((size_t)ntrain * dsub), cb;
            for (int64_t m = 0; m < m_; m++) {
                for (int64_t i = 0; i < ntrain; i++)
                    std::memcpy(&sub[i * dsub], &resid[i * dim_ + m * dsub], sizeof(float) * dsub);
                if ((s = KMeans(sub.data(), ntrain, (int)dsub, 256, 10, cb)) != Status::success) return s;
                std::memcpy(&codebooks_[(size_t)m * 256 * dsub], cb.data(), sizeof(float) * 256 * dsub);
            }
        } else if (kind_ == KNHIP_IVF_SQ8) {
            sq_trained_.assign(2 * dim_, 0.f);
            for (int64_t t = 0; t < dim_; t++) {
                float lo = FLT_MAX, hi = -FLT_MAX;
                for (int64_t i = 0; i < ntrain; i++) {
                    lo = std::min(lo, resid[i * dim_ + t]);
                    hi = std::max(hi, resid[i * dim_ + t]);
                }
                sq_trained_[t] = lo;
                sq_trained_[dim_ + t] = hi - lo;
            }
        }
        has_refine_ = cfg_.refine && (kind_ == KNHIP_IVF_PQ || kind_ == KNHIP_IVF_SQ8);
        trained_ = true;
        return Status::success;
    }

    Status Add(const DataSetPtr dataset, const Json& /*cfg*/) override {
        if (!dataset || !dataset->GetTensor()) return Status::invalid_args;
        if (kind_ != KNHIP_BRUTE_FORCE && !trained_) return Status::index_not_trained;
        if (idx_) return Status::not_implemented;  // one Add per index for now (the cuVS node's Add is a no-op)
        const int64_t rows = dataset->GetRows();
        if (dataset->GetDim() != dim_) return Status::invalid_args;
        raw_.assign((const float*)dataset->GetTensor(), (const float*)dataset->GetTensor() + rows * dim_);
        if (cosine_) NormalizeRows(raw_.data(), rows, dim_);
        knhip_desc desc{};
        desc.kind = kind_;
        desc.metric = metric_;
        desc.dim = (int32_t)dim_;
        desc.nlist = nlist_;
        desc.pq_m = (int32_t)m_;
        desc.pq_nbits = 8;
        int rc = knhip_index_create(&desc, &idx_);
        if (rc) return ToStatus(rc);
        count_ = rows;
        if (kind_ == KNHIP_BRUTE_FORCE) {
            return ToStatus(knhip_index_add_vectors(idx_, rows, raw_.data(), nullptr, 0));
        }
        std::vector<int64_t> a;
        Status s = AssignGpu(centroids_.data(), nlist_, raw_.data(), rows, (int)dim_, a);
        if (s != Status::success) return s;
        const int64_t cs = kind_ == KNHIP_IVF_FLAT ? dim_ * 4 : (kind_ == KNHIP_IVF_PQ ? m_ : dim_);
        std::vector<uint8_t> codes((size_t)rows * cs);
        if (kind_ == KNHIP_IVF_FLAT) {
            std::memcpy(codes.data(), raw_.data(), codes.size());
        } else {
            std::vector<float> resid((size_t)rows * dim_);
            for (int64_t i = 0; i < rows; i++)
                for (int64_t t = 0; t < dim_; t++) resid[i * dim_ + t] = raw_[i * dim_ + t] - centroids_[a[i] * dim_ + t];
            if (kind_ == KNHIP_IVF_PQ) {
                const int64_t dsub = dim_ / m_;
                std::vector<float> sub((size_t)rows * dsub);
                std::vector<int64_t> code;
                for (int64_t m = 0; m < m_; m++) {
                    for (int64_t i = 0; i < rows; i++)
                        std::memcpy(&sub[i * dsub], &resid[i * dim_ + m * dsub], sizeof(float) * dsub);
                    s = AssignGpu(&codebooks_[(size_t)m * 256 * dsub], 256, sub.data(), rows, (int)dsub, code);
                    if (s != Status::success) return s;
                    for (int64_t i = 0; i < rows; i++) codes[i * cs + m] = (uint8_t)code[i];
                }
            } else {  // SQ8: quantizers.h:118-133 + codecs.h:29-35
                for (int64_t i = 0; i < rows; i++)
                    for (int64_t t = 0; t < dim_; t++) {
                        const float vmin = sq_trained_[t], vdiff = sq_trained_[dim_ + t];
                        float xi = 0;
                        if (vdiff != 0) {
                            xi = (resid[i * dim_ + t] - vmin) / vdiff;
                            xi = std::min(1.0f, std::max(0.0f, xi));
                        }
                        codes[i * cs + t] = (uint8_t)(int)(255 * xi);
                    }
            }
        }
        // bucket into ArrayInvertedLists layout
        list_codes_.assign(nlist_, {});
        list_ids_.assign(nlist_, {});
        for (int64_t i = 0; i < rows; i++) {
            auto& lc = list_codes_[a[i]];
            lc.insert(lc.end(), codes.begin() + i * cs, codes.begin() + (i + 1) * cs);
            list_ids_[a[i]].push_back(i);
        }
        s = Upload();
        if (!has_refine_ && (kind_ == KNHIP_IVF_PQ || kind_ == KNHIP_IVF_SQ8)) std::vector<float>().swap(raw_);
        return s;
    }

    expected<DataSetPtr> Search(const DataSetPtr dataset, const Json& cfg, const BitsetView& bitset) const override {
        if (!idx_) return expected<DataSetPtr>::Err(Status::empty_index, "index not built");
        if (!dataset || !dataset->GetTensor()) return expected<DataSetPtr>::Err(Status::invalid_args, "null dataset");
        HipConfig c = cfg_;
        std::string msg;
        Status s = LoadConfig(cfg, c, true, &msg);
        if (s != Status::success) return expected<DataSetPtr>::Err(s, msg);
        const int64_t nq = dataset->GetRows();
        if (dataset->GetDim() != dim_) return expected<DataSetPtr>::Err(Status::invalid_args, "dim mismatch");
        const float* q = (const float*)dataset->GetTensor();
        std::vector<float> qn;
        if (cosine_) {
            qn.assign(q, q + nq * dim_);
            NormalizeRows(qn.data(), nq, dim_);
            q = qn.data();
        }
        const int64_t k = c.k;
        // every row filtered: ids -1, distances +inf, like gpu_cuvs.h:163-173
        if (!bitset.empty() && bitset.count() >= (size_t)count_) {
            auto* ids = new int64_t[nq * k];
            auto* dis = new float[nq * k];
            std::fill(ids, ids + nq * k, -1);
            std::fill(dis, dis + nq * k, std::numeric_limits<float>::infinity());
            return GenResultDataSet(nq, k, ids, dis);
        }
        // use_refine = the index carries a refine index; enabled by a search-time refine_k (ivf.cc:1080-1092)
        const bool refine = has_refine_ && cfg.contains(indexparam::REFINE_K);
        const int64_t kbase = refine ? std::min<int64_t>(1024, std::max<int64_t>(k, c.refine_k > 0 ? c.refine_k : k)) : k;
        std::unique_ptr<int64_t[]> ids(new int64_t[nq * kbase]);
        std::unique_ptr<float[]> dis(new float[nq * kbase]);
        int rc = knhip_search(idx_, q, nq, (int32_t)kbase, (int32_t)c.nprobe, bitset.empty() ? nullptr : bitset.data(),
                              (int64_t)bitset.size(), ids.get(), dis.get());
        if (rc) return expected<DataSetPtr>::Err(ToStatus(rc), knhip_last_error());
        if (refine && kbase > k) {
            // IndexRefine second stage on the host-resident raw copy (device-resident variant:
            // knhip_refine_device, used by bench.py)
            std::unique_ptr<int64_t[]> rid(new int64_t[nq * k]);
            std::unique_ptr<float[]> rdis(new float[nq * k]);
            RefineHost(q, nq, kbase, ids.get(), k, rid.get(), rdis.get());
            ids = std::move(rid);
            dis = std::move(rdis);
        }
        return GenResultDataSet(nq, k, ids.release(), dis.release());
    }

    // IvfIndexNode::RangeSearch (ivf.cc:1231-1497): radius / range_filter / max_empty_result_buckets from the
    // config, every list a candidate, results filtered to [range_filter, radius) (L2) or (radius, range_filter]
    // (IP, COSINE) (range_util.h:22-25) and returned as lims + flat arrays.
    expected<DataSetPtr> RangeSearch(const DataSetPtr dataset, const Json& cfg, const BitsetView& bitset) const override {
        if (!idx_) return expected<DataSetPtr>::Err(Status::empty_index, "index not built");
        if (!dataset || !dataset->GetTensor()) return expected<DataSetPtr>::Err(Status::invalid_args, "null dataset");
        if (dataset->GetDim() != dim_) return expected<DataSetPtr>::Err(Status::invalid_args, "dim mismatch");
        auto get_f = [&](const char* key, float dflt, float& dst) -> bool {
            dst = dflt;
            if (!cfg.contains(key)) return true;
            if (!cfg.at(key).is_number()) return false;
            dst = (float)cfg.at(key).as_double();
            return true;
        };
        float radius = 0.f, range_filter = 0.f;
        const float default_range_filter = std::numeric_limits<float>::infinity();  // config.h:583
        if (!get_f(meta::RADIUS, 0.0f, radius) || !get_f(meta::RANGE_FILTER, default_range_filter, range_filter))
            return expected<DataSetPtr>::Err(Status::type_conflict_in_json, "radius / range_filter must be numbers");
        int64_t max_empty = 2;  // ivf_config.h:53-59
        if (cfg.contains(indexparam::MAX_EMPTY_RESULT_BUCKETS)) {
            if (!cfg.at(indexparam::MAX_EMPTY_RESULT_BUCKETS).is_number())
                return expected<DataSetPtr>::Err(Status::type_conflict_in_json, "max_empty_result_buckets");
            max_empty = cfg.at(indexparam::MAX_EMPTY_RESULT_BUCKETS).as_int();
            if (max_empty < 0 || max_empty > 65536)
                return expected<DataSetPtr>::Err(Status::out_of_range_in_json, "max_empty_result_buckets");
        }
        const int64_t nq = dataset->GetRows();
        const float* q = (const float*)dataset->GetTensor();
        std::vector<float> qn;
        if (cosine_) {
            qn.assign(q, q + nq * dim_);
            NormalizeRows(qn.data(), nq, dim_);
            q = qn.data();
        }
        std::vector<int64_t> lims(nq + 1);
        int64_t* ids = nullptr;
        float* dis = nullptr;
        int rc = knhip_range_search(idx_, q, nq, radius, (int32_t)max_empty, bitset.empty() ? nullptr : bitset.data(),
                                    (int64_t)bitset.size(), lims.data(), &ids, &dis);
        if (rc) return expected<DataSetPtr>::Err(ToStatus(rc), knhip_last_error());
        const bool is_ip = metric_ != KNHIP_L2;
        auto* out_lims = new size_t[nq + 1];
        auto* out_ids = new int64_t[std::max<int64_t>(lims[nq], 1)];
        auto* out_dis = new float[std::max<int64_t>(lims[nq], 1)];
        size_t n = 0;
        out_lims[0] = 0;
        for (int64_t i = 0; i < nq; i++) {
            for (int64_t j = lims[i]; j < lims[i + 1]; j++) {
                const float v = dis[j];
                const bool keep = range_filter == default_range_filter ||
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
        return GenRangeResultDataSet(nq, out_lims, out_ids, out_dis);
    }
    expected<DataSetPtr> GetVectorByIds(const DataSetPtr dataset) const override {
        if (!HasRawData(cfg_.metric) || raw_.empty())
            return expected<DataSetPtr>::Err(Status::not_implemented, "no raw data");
        const int64_t n = dataset->GetRows();
        const int64_t* ids = dataset->GetIds();
        auto* out = new float[n * dim_];
        for (int64_t i = 0; i < n; i++) {
            if (ids[i] < 0 || ids[i] >= count_) {
                delete[] out;
                return expected<DataSetPtr>::Err(Status::invalid_args, "id out of range");
            }
            std::memcpy(out + i * dim_, &raw_[ids[i] * dim_], sizeof(float) * dim_);
        }
        auto ds = GenDataSet(n, dim_, out);
        ds->SetIsOwner(true);
        return ds;
    }
    bool HasRawData(const std::string& metric_type) const override {
        // cosine stores normalised vectors (ivf.cc StaticHasRawData semantics)
        return (kind_ == KNHIP_BRUTE_FORCE || kind_ == KNHIP_IVF_FLAT) && metric_type != metric::COSINE;
    }
    expected<DataSetPtr> GetIndexMeta(const Json&) const override {
        return expected<DataSetPtr>::Err(Status::not_implemented, "GetIndexMeta not implemented");
    }

    // One named blob in the BinarySet (ivf.cc:1717-1744), holding the FAISS byte format the CPU nodes
    // write (faiss_io.h): IxF2/IxFI, IwFl, IwSq, IwPQ, wrapped in IxRF when built with refine.
    Status Serialize(BinarySet& binset) const override {
        if (!idx_) return Status::empty_index;
        using namespace knhip_host;
        FaissIndexData x;
        auto fill_hdr = [&](FaissHeader& h, int64_t ntotal) {
            h.d = (int32_t)dim_;
            h.ntotal = ntotal;
            h.dummy[0] = cosine_ ? 1 : 0;  // Knowhere's is_cosine byte (cppcontrib/knowhere/impl/index_write.cpp:84-88)
            h.is_trained = true;
            h.metric = metric_ == KNHIP_L2 ? 1 : 0;
        };
        const uint32_t flat_cc = metric_ == KNHIP_L2 ? FourCC("IxF2") : FourCC("IxFI");
        fill_hdr(x.hdr, count_);
        if (kind_ == KNHIP_BRUTE_FORCE) {
            x.fourcc = flat_cc;
            x.xb = raw_;
        } else {
            x.fourcc = kind_ == KNHIP_IVF_FLAT ? FourCC("IwFl") : kind_ == KNHIP_IVF_PQ ? FourCC("IwPQ") : FourCC("IwSq");
            x.nlist = (uint64_t)nlist_;
            x.nprobe = (uint64_t)cfg_.nprobe;
            x.quantizer.fourcc = flat_cc;
            fill_hdr(x.quantizer.hdr, nlist_);
            x.quantizer.hdr.dummy[0] = 0;
            x.quantizer.xb = centroids_;
            x.by_residual = true;
            x.code_size = (uint64_t)CodeSize();
            if (kind_ == KNHIP_IVF_PQ) {
                x.pq_d = (uint64_t)dim_; x.pq_M = (uint64_t)m_; x.pq_nbits = 8;
                x.pq_centroids = codebooks_;
            } else if (kind_ == KNHIP_IVF_SQ8) {
                x.sq_qtype = 0;      // ScalarQuantizer::QT_8bit
                x.sq_rangestat = 0;  // RS_minmax
                x.sq_d = (uint64_t)dim_; x.sq_code_size = (uint64_t)dim_;
                x.sq_trained = sq_trained_;
            }
            x.codes = list_codes_;
            x.ids = list_ids_;
            size_t non0 = 0;
            for (auto& l : list_ids_) non0 += !l.empty();
            x.lists_sparse = !(non0 > (size_t)nlist_ / 2);  // index_write.cpp:309-316
            if (kind_ == KNHIP_IVF_FLAT && cosine_) {       // Knowhere cosine IVF-Flat carries the row norms
                x.with_norm = true;
                x.norms.assign(nlist_, {});
                for (int64_t l = 0; l < nlist_; l++)
                    for (size_t i = 0; i < list_ids_[l].size(); i++) {
                        const float* v = (const float*)&list_codes_[l][i * dim_ * 4];
                        float s = 0;
                        for (int64_t t = 0; t < dim_; t++) s += v[t] * v[t];
                        x.norms[l].push_back(std::sqrt(s));
                    }
            }
            if (has_refine_) {  // IndexRefineFlat (ivf.cc:673-700)
                x.has_refine = true;
                fill_hdr(x.refine_hdr, count_);
                x.refine_index.fourcc = flat_cc;
                fill_hdr(x.refine_index.hdr, count_);
                x.refine_index.hdr.dummy[0] = 0;
                x.refine_index.xb = raw_;
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

    // Accepts the blob of this node AND of the CPU node of the same kind (FLAT / IVF_FLAT / IVF_PQ /
    // IVF_SQ8, or the knowhere-1.x name "IVF", ivf.cc:1750-1757): same bytes.
    Status Deserialize(const BinarySet& binset, const Json& cfg) override {
        using namespace knhip_host;
        static const char* cpu_names[] = {"FLAT", "IVF_FLAT", "IVF_PQ", "IVF_SQ8"};
        BinaryPtr b = binset.GetByName(Type());
        if (!b) b = binset.GetByName(cpu_names[kind_ == KNHIP_BRUTE_FORCE ? 0 : kind_ == KNHIP_IVF_FLAT ? 1
                                               : kind_ == KNHIP_IVF_PQ    ? 2 : 3]);
        if (!b && kind_ != KNHIP_BRUTE_FORCE) b = binset.GetByName("IVF");
        if (!b) return Status::invalid_binary_set;
        FaissIndexData x;
        std::string err;
        if (!ParseFaissIndex(b->data.get(), (size_t)b->size, &x, &err)) return Status::invalid_serialized_index_type;
        const bool flat = x.fourcc == FourCC("IxF2") || x.fourcc == FourCC("IxFI");
        const bool kind_ok = (kind_ == KNHIP_BRUTE_FORCE && flat) || (kind_ == KNHIP_IVF_FLAT && x.fourcc == FourCC("IwFl")) ||
                             (kind_ == KNHIP_IVF_PQ && x.fourcc == FourCC("IwPQ")) ||
                             (kind_ == KNHIP_IVF_SQ8 && x.fourcc == FourCC("IwSq"));
        if (!kind_ok) return Status::invalid_serialized_index_type;
        if (x.hdr.metric != 0 && x.hdr.metric != 1) return Status::invalid_metric_type;
        if (!flat && x.quantizer.hdr.metric != x.hdr.metric) return Status::not_implemented;
        if (kind_ == KNHIP_IVF_PQ &&
            (x.pq_nbits != 8 || !x.by_residual || !(x.pq_M == 8 || x.pq_M == 16 || x.pq_M == 32 || x.pq_M == 64)))
            return Status::not_implemented;
        if (kind_ == KNHIP_IVF_SQ8 && (x.sq_qtype != 0 || !x.by_residual || x.sq_trained.size() != 2 * (size_t)x.hdr.d))
            return Status::not_implemented;
        metric_ = x.hdr.metric == 1 ? KNHIP_L2 : KNHIP_IP;
        cosine_ = x.hdr.is_cosine();
        if (cfg.contains(meta::METRIC_TYPE) && cfg.at(meta::METRIC_TYPE).is_string())
            cosine_ = cosine_ || cfg.at(meta::METRIC_TYPE).as_string() == metric::COSINE;
        cfg_.metric = cosine_ ? metric::COSINE : (metric_ == KNHIP_L2 ? metric::L2 : metric::IP);
        dim_ = x.hdr.d;
        count_ = x.hdr.ntotal;
        nlist_ = (int64_t)x.nlist;
        if (x.nprobe >= 1 && x.nprobe <= 65536) cfg_.nprobe = (int64_t)x.nprobe;  // the index's default nprobe
        m_ = (int64_t)x.pq_M;
        centroids_ = std::move(x.quantizer.xb);
        codebooks_ = std::move(x.pq_centroids);
        sq_trained_ = std::move(x.sq_trained);
        list_codes_ = std::move(x.codes);
        list_ids_ = std::move(x.ids);
        has_refine_ = x.has_refine;
        raw_.clear();
        if (kind_ == KNHIP_BRUTE_FORCE) {
            raw_ = std::move(x.xb);
        } else if (x.has_refine) {
            if (x.refine_index.hdr.ntotal != count_) return Status::invalid_serialized_index_type;
            raw_ = std::move(x.refine_index.xb);
        } else if (kind_ == KNHIP_IVF_FLAT) {  // raw rows back in id order (make_direct_map, ivf.cc:1815-1828)
            raw_.assign((size_t)count_ * dim_, 0.f);
            for (int64_t l = 0; l < nlist_; l++)
                for (size_t i = 0; i < list_ids_[l].size(); i++) {
                    const int64_t id = list_ids_[l][i];
                    if (id < 0 || id >= count_) { raw_.clear(); l = nlist_; break; }
                    std::memcpy(&raw_[id * dim_], &list_codes_[l][i * dim_ * 4], sizeof(float) * dim_);
                }
        }
        knhip_index_destroy(idx_);
        idx_ = nullptr;
        knhip_desc desc{};
        desc.kind = kind_; desc.metric = metric_; desc.dim = (int32_t)dim_; desc.nlist = nlist_;
        desc.pq_m = (int32_t)m_; desc.pq_nbits = 8;
        int rc = knhip_index_create(&desc, &idx_);
        if (rc) return ToStatus(rc);
        trained_ = true;
        if (kind_ == KNHIP_BRUTE_FORCE) return ToStatus(knhip_index_add_vectors(idx_, count_, raw_.data(), nullptr, 0));
        return Upload();
    }
    Status DeserializeFromFile(const std::string&, const Json&) override {
        return Status::not_implemented;  // as the cuVS node (gpu_cuvs.h:250-253)
    }

    int64_t Dim() const override { return dim_; }
    int64_t Size() const override { return idx_ ? knhip_index_device_bytes(idx_) : 0; }
    int64_t Count() const override { return count_; }
    std::string Type() const override {
        switch (kind_) {
            case KNHIP_BRUTE_FORCE: return IndexEnum::INDEX_HIP_BRUTEFORCE;
            case KNHIP_IVF_FLAT: return IndexEnum::INDEX_HIP_IVFFLAT;
            case KNHIP_IVF_PQ: return IndexEnum::INDEX_HIP_IVFPQ;
            default: return IndexEnum::INDEX_HIP_IVFSQ8;
        }
    }

 private:
    int64_t CodeSize() const { return kind_ == KNHIP_IVF_FLAT ? dim_ * 4 : (kind_ == KNHIP_IVF_PQ ? m_ : dim_); }

    Status Upload() {
        int rc = knhip_index_set_coarse(idx_, centroids_.data());
        if (rc) return ToStatus(rc);
        if (kind_ == KNHIP_IVF_PQ && (rc = knhip_index_set_pq(idx_, codebooks_.data()))) return ToStatus(rc);
        if (kind_ == KNHIP_IVF_SQ8 &&
            (rc = knhip_index_set_sq(idx_, sq_trained_.data(), sq_trained_.data() + dim_)))
            return ToStatus(rc);
        std::vector<int64_t> sizes(nlist_);
        std::vector<const uint8_t*> cp(nlist_);
        std::vector<const int64_t*> ip(nlist_);
        for (int64_t l = 0; l < nlist_; l++) {
            sizes[l] = (int64_t)list_ids_[l].size();
            cp[l] = list_codes_[l].data();
            ip[l] = list_ids_[l].data();
        }
        return ToStatus(knhip_index_add_lists(idx_, sizes.data(), cp.data(), ip.data()));
    }

    // exact re-rank in the reference's scalar order (IndexRefine.cpp:108-140)
    void RefineHost(const float* q, int64_t nq, int64_t kbase, const int64_t* cand, int64_t k, int64_t* oi,
                    float* od) const {
        const bool l2 = metric_ == KNHIP_L2;
        std::vector<std::pair<float, int64_t>> v;
        for (int64_t i = 0; i < nq; i++) {
            v.clear();
            for (int64_t j = 0; j < kbase; j++) {
                const int64_t id = cand[i * kbase + j];
                if (id < 0) break;
                const float* y = &raw_[id * dim_];
                const float* x = q + i * dim_;
                float acc = 0;
                for (int64_t t = 0; t < dim_; t++) {
                    if (l2) {
                        const float d = x[t] - y[t];
                        acc += d * d;
                    } else {
                        acc += x[t] * y[t];
                    }
                }
                v.emplace_back(acc, id);
            }
            std::sort(v.begin(), v.end(), [l2](const auto& a, const auto& b) {
                return l2 ? (a.first < b.first || (a.first == b.first && a.second < b.second))
                          : (a.first > b.first || (a.first == b.first && a.second > b.second));
            });
            for (int64_t j = 0; j < k; j++) {
                oi[i * k + j] = j < (int64_t)v.size() ? v[j].second : -1;
                od[i * k + j] = j < (int64_t)v.size() ? v[j].first : (l2 ? FLT_MAX : -FLT_MAX);
            }
        }
    }

    int kind_;
    int metric_ = KNHIP_L2;
    bool cosine_ = false, trained_ = false, has_refine_ = false;
    int64_t dim_ = 0, nlist_ = 0, m_ = 0, count_ = 0;
    HipConfig cfg_;
    knhip_index* idx_ = nullptr;
    std::vector<float> centroids_, codebooks_, sq_trained_, raw_;
    std::vector<std::vector<uint8_t>> list_codes_;
    std::vector<std::vector<int64_t>> list_ids_;
};

// static-init registration, as every node's translation unit does (index_factory.h:75-77)
KNOWHERE_HIP_REGISTER_GLOBAL(GPU_HIP_BRUTE_FORCE, HipIndexNode, KNHIP_BRUTE_FORCE);
KNOWHERE_HIP_REGISTER_GLOBAL(GPU_HIP_IVF_FLAT, HipIndexNode, KNHIP_IVF_FLAT);
KNOWHERE_HIP_REGISTER_GLOBAL(GPU_HIP_IVF_PQ, HipIndexNode, KNHIP_IVF_PQ);
KNOWHERE_HIP_REGISTER_GLOBAL(GPU_HIP_IVF_SQ8, HipIndexNode, KNHIP_IVF_SQ8);

// BruteForce::Search<fp32> (include/knowhere/comp/brute_force.h:27-31) through the same kernels
template <>
expected<DataSetPtr> BruteForce::Search<fp32>(const DataSetPtr base, const DataSetPtr query, const Json& config,
                                              const BitsetView& bitset) {
    HipIndexNode node(Version::GetCurrentVersion(), KNHIP_BRUTE_FORCE);
    Status s = node.Build(base, config);
    if (s != Status::success) return expected<DataSetPtr>::Err(s, "brute force build failed");
    return node.Search(query, config, bitset);
}

}  // namespace knowhere
