// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin C-ABI over the REAL reference arithmetic: the baseline FAISS objects vendored in the
// reference tree (/root/reference/thirdparty/faiss/faiss), compiled from where they lie by
// oracle/Makefile into oracle/_ref/libknowhere_ref.so.  Nothing here re-implements a
// distance; it only drives the reference the way Knowhere does:
//   * IvfIndexNode::Search  (reference src/index/ivf/ivf.cc:915-1159): one task per query,
//     index->search(1, q, k, ...), OpenMP pinned to 1 thread inside the task,
//     IVFSearchParameters{nprobe, max_codes=0, sel = BitsetViewIDSelector | nullptr}.
//   * FlatIndexNode::Search (reference src/index/flat/flat.cc:76-148): IndexFlat::search(1,..)
//   * BitsetViewIDSelector::is_member(id) = !bitset.test(id), LSB-first bytes
//     (reference include/knowhere/bitsetview_idselector.h:20-31, index_node.h:646).
// Used by tests/ to pin oracle.c (the plain-C restatement) and, as "kind":"reference",
// by bench.py's cpu_baseline leg.

#include <faiss/Clustering.h>
#include <faiss/IndexFlat.h>
#include <faiss/IndexIVF.h>
#include <faiss/IndexIVFFlat.h>
#include <faiss/IndexIVFPQ.h>
#include <faiss/IndexRefine.h>
#include <faiss/IndexScalarQuantizer.h>
#include <faiss/impl/IDSelector.h>
#include <faiss/impl/io.h>
#include <faiss/index_io.h>
#include <faiss/invlists/InvertedLists.h>
#include <faiss/utils/distances.h>

#include <omp.h>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include <cstdlib>

namespace {

struct BitsetSelector : faiss::IDSelector {
    const uint8_t* bits;
    int64_t nbits;
    BitsetSelector(const uint8_t* b, int64_t n) : bits(b), nbits(n) {}
    bool is_member(faiss::idx_t id) const override {
        if (id < 0 || id >= nbits) {
            return true;
        }
        return !((bits[id >> 3] >> (id & 7)) & 1);
    }
};

enum Kind { K_FLAT = 0, K_IVF_FLAT = 1, K_IVF_PQ = 2, K_IVF_SQ8 = 3 };

struct RefIndex {
    int kind = 0, metric = 0, d = 0;
    std::unique_ptr<faiss::IndexFlat> quantizer;  // coarse quantizer (IVF kinds)
    std::unique_ptr<faiss::Index> index;          // the searched index
    faiss::IndexIVF* ivf() const {
        return dynamic_cast<faiss::IndexIVF*>(index.get());
    }
};

thread_local std::string g_err;

template <class F>
int guarded(F&& f) {
    try {
        f();
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

}  // namespace

extern "C" {

const char* ref_last_error() {
    return g_err.c_str();
}

void* ref_create(int kind, int metric, int d, int nlist, int M, int nbits) {
    auto* h = new RefIndex();
    h->kind = kind;
    h->metric = metric;
    h->d = d;
    faiss::MetricType mt = metric == 0 ? faiss::METRIC_L2 : faiss::METRIC_INNER_PRODUCT;
    int rc = guarded([&] {
        if (kind == K_FLAT) {
            h->index.reset(new faiss::IndexFlat(d, mt));
            return;
        }
        h->quantizer.reset(new faiss::IndexFlat(d, mt));
        if (kind == K_IVF_FLAT) {
            h->index.reset(new faiss::IndexIVFFlat(h->quantizer.get(), d, nlist, mt));
        } else if (kind == K_IVF_PQ) {
            h->index.reset(new faiss::IndexIVFPQ(h->quantizer.get(), d, nlist, M, nbits, mt));
        } else if (kind == K_IVF_SQ8) {
            h->index.reset(new faiss::IndexIVFScalarQuantizer(
                    h->quantizer.get(), d, nlist, faiss::ScalarQuantizer::QT_8bit, mt, true));
        } else {
            throw std::runtime_error("bad kind");
        }
    });
    if (rc != 0) {
        delete h;
        return nullptr;
    }
    return h;
}

void ref_destroy(void* hv) {
    auto* h = static_cast<RefIndex*>(hv);
    if (!h) {
        return;
    }
    h->index.reset();  // index does not own the quantizer (own_fields=false)
    h->quantizer.reset();
    delete h;
}

// the reference's own k-means (faiss::Clustering with an IndexFlat of the metric as the assigner): pins oracle.c's
// orc_kmeans.  Bit-reproducible only while the assigner stays on its sequential path (fewer than
// distance_compute_blas_threshold = 20 training points), which is what the pin test uses.
int ref_kmeans(int metric, int d, int64_t n, const float* x, int64_t k, int niter, int max_points, int64_t seed,
               int spherical, float* centroids) {
    return guarded([&] {
        faiss::ClusteringParameters cp;
        cp.niter = niter;
        cp.max_points_per_centroid = max_points;
        cp.min_points_per_centroid = 1;
        cp.seed = (int)seed;
        cp.spherical = spherical != 0;
        faiss::Clustering clus(d, (int)k, cp);
        faiss::IndexFlat index(d, metric == 0 ? faiss::METRIC_L2 : faiss::METRIC_INNER_PRODUCT);
        clus.train(n, x, index);
        std::memcpy(centroids, clus.centroids.data(), sizeof(float) * (size_t)k * d);
    });
}

int ref_train(void* hv, int64_t n, const float* x) {
    auto* h = static_cast<RefIndex*>(hv);
    return guarded([&] { h->index->train(n, x); });
}

int ref_add(void* hv, int64_t n, const float* x, const int64_t* ids) {
    auto* h = static_cast<RefIndex*>(hv);
    return guarded([&] {
        if (ids && h->kind != K_FLAT) {
            h->index->add_with_ids(n, x, ids);
        } else {
            h->index->add(n, x);
        }
    });
}

int64_t ref_ntotal(void* hv) {
    return static_cast<RefIndex*>(hv)->index->ntotal;
}

int64_t ref_nlist(void* hv) {
    auto* ivf = static_cast<RefIndex*>(hv)->ivf();
    return ivf ? (int64_t)ivf->nlist : 0;
}

int64_t ref_code_size(void* hv) {
    auto* ivf = static_cast<RefIndex*>(hv)->ivf();
    return ivf ? (int64_t)ivf->code_size : 0;
}

/* ---------------- export (reference-trained index -> plain arrays) ---------------- */

int ref_get_flat_vectors(void* hv, float* out) {
    auto* h = static_cast<RefIndex*>(hv);
    auto* f = dynamic_cast<faiss::IndexFlat*>(h->index.get());
    if (!f) {
        return -1;
    }
    std::memcpy(out, f->get_xb(), sizeof(float) * f->ntotal * f->d);
    return 0;
}

int ref_get_centroids(void* hv, float* out) {
    auto* h = static_cast<RefIndex*>(hv);
    if (!h->quantizer) {
        return -1;
    }
    std::memcpy(out, h->quantizer->get_xb(), sizeof(float) * h->quantizer->ntotal * h->d);
    return 0;
}

int ref_get_pq_centroids(void* hv, float* out) {
    auto* p = dynamic_cast<faiss::IndexIVFPQ*>(static_cast<RefIndex*>(hv)->index.get());
    if (!p) {
        return -1;
    }
    std::memcpy(out, p->pq.centroids.data(), sizeof(float) * p->pq.centroids.size());
    return 0;
}

int ref_get_sq_trained(void* hv, float* out) {
    auto* p = dynamic_cast<faiss::IndexIVFScalarQuantizer*>(static_cast<RefIndex*>(hv)->index.get());
    if (!p) {
        return -1;
    }
    std::memcpy(out, p->sq.trained.data(), sizeof(float) * p->sq.trained.size());
    return 0;
}

int ref_use_precomputed_table(void* hv) {
    auto* p = dynamic_cast<faiss::IndexIVFPQ*>(static_cast<RefIndex*>(hv)->index.get());
    return p ? p->use_precomputed_table : -2;
}

int64_t ref_get_precomputed_table(void* hv, float* out) {
    auto* p = dynamic_cast<faiss::IndexIVFPQ*>(static_cast<RefIndex*>(hv)->index.get());
    if (!p) {
        return -1;
    }
    if (out) {
        std::memcpy(out, p->precomputed_table.data(), sizeof(float) * p->precomputed_table.size());
    }
    return (int64_t)p->precomputed_table.size();
}

int64_t ref_list_size(void* hv, int64_t l) {
    auto* ivf = static_cast<RefIndex*>(hv)->ivf();
    return ivf ? (int64_t)ivf->invlists->list_size(l) : -1;
}

int ref_get_list(void* hv, int64_t l, uint8_t* codes, int64_t* ids) {
    auto* ivf = static_cast<RefIndex*>(hv)->ivf();
    if (!ivf) {
        return -1;
    }
    size_t n = ivf->invlists->list_size(l);
    if (n == 0) {
        return 0;
    }
    faiss::InvertedLists::ScopedCodes sc(ivf->invlists, l);
    faiss::InvertedLists::ScopedIds si(ivf->invlists, l);
    std::memcpy(codes, sc.get(), n * ivf->code_size);
    std::memcpy(ids, si.get(), n * sizeof(int64_t));
    return 0;
}

/* ---------------- import (plain arrays -> reference index, no training) ---------------- */

int ref_set_centroids(void* hv, int64_t nlist, const float* c) {
    auto* h = static_cast<RefIndex*>(hv);
    return guarded([&] {
        h->quantizer->reset();
        h->quantizer->add(nlist, c);
        auto* ivf = h->ivf();
        ivf->is_trained = true;  // encoder params are set separately
    });
}

int ref_set_pq_centroids(void* hv, const float* cb) {
    auto* p = dynamic_cast<faiss::IndexIVFPQ*>(static_cast<RefIndex*>(hv)->index.get());
    if (!p) {
        return -1;
    }
    std::memcpy(p->pq.centroids.data(), cb, sizeof(float) * p->pq.centroids.size());
    return 0;
}

int ref_set_sq_trained(void* hv, const float* t) {
    auto* p = dynamic_cast<faiss::IndexIVFScalarQuantizer*>(static_cast<RefIndex*>(hv)->index.get());
    if (!p) {
        return -1;
    }
    p->sq.trained.assign(t, t + 2 * p->d);
    return 0;
}

int ref_add_list_entries(void* hv, int64_t l, int64_t n, const uint8_t* codes, const int64_t* ids) {
    auto* h = static_cast<RefIndex*>(hv);
    return guarded([&] {
        auto* ivf = h->ivf();
        ivf->invlists->add_entries(l, n, ids, codes);
        ivf->ntotal += n;
    });
}

/// (re)build the IVFPQ precomputed table exactly as IndexIVFPQ::train_encoder does
/// (reference thirdparty/faiss/faiss/IndexIVFPQ.cpp:515-523).
int ref_finalize(void* hv) {
    auto* h = static_cast<RefIndex*>(hv);
    return guarded([&] {
        if (auto* p = dynamic_cast<faiss::IndexIVFPQ*>(h->index.get())) {
            p->use_precomputed_table = 0;
            p->precompute_table();
        }
    });
}

/* ---------------- search, driven the way Knowhere drives it ---------------- */

int ref_search(
        void* hv,
        int64_t nq,
        const float* q,
        int64_t k,
        int64_t nprobe,
        const uint8_t* bitset,
        int64_t nbits,
        float* D,
        int64_t* I,
        int nthreads) {
    auto* h = static_cast<RefIndex*>(hv);
    std::string err;
    int failed = 0;
    if (nthreads < 1) {
        nthreads = 1;
    }
    omp_set_max_active_levels(1);  // inner faiss omp regions run single-threaded
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
    for (int64_t i = 0; i < nq; i++) {
        try {
            // one thread inside each task (Knowhere: omp = 1 inside a search task, ivf.cc:920).  Without this the flat
            // index's own `omp parallel num_threads(min(nx, omp_get_max_threads()))` returned untouched result arrays when
            // this loop ran on more than one thread (seen with libgomp in the dev container: ids all 0)
            omp_set_num_threads(1);
            std::unique_ptr<BitsetSelector> sel;
            if (bitset) {
                sel.reset(new BitsetSelector(bitset, nbits));
            }
            if (h->kind == K_FLAT) {
                faiss::SearchParameters sp;
                sp.sel = sel.get();
                h->index->search(1, q + i * h->d, k, D + i * k, I + i * k, &sp);
            } else {
                faiss::IVFSearchParameters sp;
                sp.nprobe = nprobe;
                sp.max_codes = 0;
                sp.sel = sel.get();
                h->index->search(1, q + i * h->d, k, D + i * k, I + i * k, &sp);
            }
        } catch (const std::exception& e) {
#pragma omp critical
            {
                failed = 1;
                err = e.what();
            }
        }
    }
    if (failed) {
        g_err = err;
        return -1;
    }
    return 0;
}

/// Knowhere's refine path: IndexRefine over the IVF index with a flat fp32 refine index
/// (reference src/index/ivf/ivf.cc:1073-1103, src/index/refine/refine_utils.cc:99).
int ref_search_refine(
        void* hv,
        int64_t nb,
        const float* xb,
        int64_t nq,
        const float* q,
        int64_t k,
        float k_factor,
        int64_t nprobe,
        float* D,
        int64_t* I) {
    auto* h = static_cast<RefIndex*>(hv);
    return guarded([&] {
        faiss::MetricType mt = h->metric == 0 ? faiss::METRIC_L2 : faiss::METRIC_INNER_PRODUCT;
        faiss::IndexFlat flat(h->d, mt);
        flat.add(nb, xb);
        faiss::IndexRefine refine(h->index.get(), &flat);
        refine.ntotal = h->index->ntotal; // both were populated independently
        faiss::IVFSearchParameters ivfp;
        ivfp.nprobe = nprobe;
        faiss::IndexRefineSearchParameters rp;
        rp.k_factor = k_factor;
        rp.base_index_params = &ivfp;
        for (int64_t i = 0; i < nq; i++) {
            refine.search(1, q + i * h->d, k, D + i * k, I + i * k, &rp);
        }
    });
}

/// what Knowhere does right after constructing the refine quantizer (reference src/index/refine/refine_utils.cc:176-180):
/// QT_4bit_uniform with L2 takes its one range from the 1 % / 99 % quantiles
static void knowhere_refine_sq_setup(faiss::IndexScalarQuantizer& sq, int row_type, bool is_l2) {
    if (row_type == 6 && is_l2) {
        sq.sq.rangestat = faiss::ScalarQuantizer::RS_quantiles;
        sq.sq.rangestat_arg = 0.01;
    }
}

static faiss::ScalarQuantizer::QuantizerType row_qtype(int row_type) {
    switch (row_type) {
        case 1: return faiss::ScalarQuantizer::QT_fp16;
        case 2: return faiss::ScalarQuantizer::QT_bf16;
        case 3: return faiss::ScalarQuantizer::QT_8bit;
        case 4: return faiss::ScalarQuantizer::QT_6bit;
        case 5: return faiss::ScalarQuantizer::QT_8bit_direct_signed;
        case 6: return faiss::ScalarQuantizer::QT_4bit_uniform;
        default: throw std::runtime_error("row type: 1 fp16, 2 bf16, 3 sq8, 4 sq6, 5 int8, 6 sq4u");
    }
}

/// The refine index Knowhere builds for refine_type = fp16 / bf16 / sq8: faiss::IndexScalarQuantizer(d, qtype, metric),
/// trained and filled with the raw vectors (reference src/index/refine/refine_utils.cc:150-185).  codes_out
/// [nb][code_size] and trained_out (2*d floats for sq8) receive its state if non-null.
int ref_sq_rows(int row_type, int metric, int d, int64_t nb, const float* xb, uint8_t* codes_out, float* trained_out) {
    return guarded([&] {
        faiss::IndexScalarQuantizer sq(d, row_qtype(row_type), metric == 0 ? faiss::METRIC_L2 : faiss::METRIC_INNER_PRODUCT);
        knowhere_refine_sq_setup(sq, row_type, metric == 0);
        sq.train(nb, xb);
        sq.add(nb, xb);
        if (codes_out) {
            std::memcpy(codes_out, sq.codes.data(), sq.codes.size());
        }
        if (trained_out && !sq.sq.trained.empty()) {
            std::memcpy(trained_out, sq.sq.trained.data(), sizeof(float) * sq.sq.trained.size());
        }
    });
}

/// IndexRefine(base = h, refine = IndexScalarQuantizer(xb)).search, one query per call
int ref_search_refine_sq(
        void* hv,
        int row_type,
        int64_t nb,
        const float* xb,
        int64_t nq,
        const float* q,
        int64_t k,
        float k_factor,
        int64_t nprobe,
        float* D,
        int64_t* I) {
    auto* h = static_cast<RefIndex*>(hv);
    return guarded([&] {
        faiss::MetricType mt = h->metric == 0 ? faiss::METRIC_L2 : faiss::METRIC_INNER_PRODUCT;
        faiss::IndexScalarQuantizer sq(h->d, row_qtype(row_type), mt);
        knowhere_refine_sq_setup(sq, row_type, mt == faiss::METRIC_L2);
        sq.train(nb, xb);
        sq.add(nb, xb);
        faiss::IndexRefine refine(h->index.get(), &sq);
        refine.ntotal = h->index->ntotal;
        faiss::IVFSearchParameters ivfp;
        ivfp.nprobe = nprobe;
        faiss::IndexRefineSearchParameters rp;
        rp.k_factor = k_factor;
        rp.base_index_params = &ivfp;
        for (int64_t i = 0; i < nq; i++) {
            refine.search(1, q + i * h->d, k, D + i * k, I + i * k, &rp);
        }
    });
}

/// write_index(IndexRefine(base, IndexScalarQuantizer(xb))) -- what the node serialises with a quantised refine_type
int64_t ref_serialize_sq(void* hv, int row_type, int64_t nb, const float* xb, uint8_t* out, int64_t cap) {
    auto* h = static_cast<RefIndex*>(hv);
    int64_t n = -1;
    int rc = guarded([&] {
        faiss::MetricType mt = h->metric == 0 ? faiss::METRIC_L2 : faiss::METRIC_INNER_PRODUCT;
        faiss::IndexScalarQuantizer sq(h->d, row_qtype(row_type), mt);
        knowhere_refine_sq_setup(sq, row_type, mt == faiss::METRIC_L2);
        sq.train(nb, xb);
        sq.add(nb, xb);
        faiss::IndexRefine refine(h->index.get(), &sq);
        refine.ntotal = h->index->ntotal;
        faiss::VectorIOWriter w;
        faiss::write_index(&refine, &w);
        n = (int64_t)w.data.size();
        if (n <= cap) {
            std::memcpy(out, w.data.data(), w.data.size());
        }
    });
    return rc == 0 ? n : -1;
}

/// read_index of ANY serialized IndexRefine (bytes written by the HIP node) and search through it, one query per call
int ref_blob_search_refine(
        const uint8_t* data,
        int64_t size,
        int64_t nq,
        const float* q,
        int64_t k,
        float k_factor,
        int64_t nprobe,
        float* D,
        int64_t* I) {
    return guarded([&] {
        faiss::VectorIOReader r;
        r.data.assign(data, data + size);
        std::unique_ptr<faiss::Index> idx(faiss::read_index(&r));
        auto* rf = dynamic_cast<faiss::IndexRefine*>(idx.get());
        if (!rf) {
            throw std::runtime_error("not an IndexRefine");
        }
        faiss::IVFSearchParameters ivfp;
        ivfp.nprobe = nprobe;
        faiss::IndexRefineSearchParameters rp;
        rp.k_factor = k_factor;
        rp.base_index_params = &ivfp;
        for (int64_t i = 0; i < nq; i++) {
            rf->search(1, q + i * idx->d, k, D + i * k, I + i * k, &rp);
        }
    });
}

/// faiss::write_index into memory -- the bytes IvfIndexNode::SerializeImpl puts into the BinarySet
/// (reference src/index/ivf/ivf.cc:1717-1744).  With nb_refine > 0 the index is first wrapped the
/// way Knowhere's `refine` build option does: IndexRefine(base, IndexFlat(raw vectors))
/// (ivf.cc:673-700).  Returns the byte count (also when cap is too small), -1 on error.
int64_t ref_serialize(void* hv, int64_t nb_refine, const float* xb_refine, uint8_t* out, int64_t cap) {
    auto* h = static_cast<RefIndex*>(hv);
    int64_t n = -1;
    int rc = guarded([&] {
        faiss::VectorIOWriter w;
        if (nb_refine > 0) {
            faiss::MetricType mt = h->metric == 0 ? faiss::METRIC_L2 : faiss::METRIC_INNER_PRODUCT;
            faiss::IndexFlat flat(h->d, mt);
            flat.add(nb_refine, xb_refine);
            faiss::IndexRefine refine(h->index.get(), &flat);
            refine.ntotal = h->index->ntotal;
            faiss::write_index(&refine, &w);
        } else {
            faiss::write_index(h->index.get(), &w);
        }
        n = (int64_t)w.data.size();
        if (n <= cap) {
            std::memcpy(out, w.data.data(), w.data.size());
        }
    });
    return rc == 0 ? n : -1;
}

/// faiss::read_index from memory (IvfIndexNode::Deserialize, ivf.cc:1750-1834).  An IndexRefine
/// wrapper is unwrapped to its base index; *nb_refine / xb_refine (if non-null, capacity ntotal*d)
/// receive the refine index's vectors.  Returns a handle usable with ref_search / ref_destroy.
void* ref_deserialize(const uint8_t* data, int64_t size, int64_t* nb_refine, float* xb_refine) {
    auto* h = new RefIndex();
    int rc = guarded([&] {
        faiss::VectorIOReader r;
        r.data.assign(data, data + size);
        std::unique_ptr<faiss::Index> idx(faiss::read_index(&r));
        if (nb_refine) {
            *nb_refine = 0;
        }
        if (auto* rf = dynamic_cast<faiss::IndexRefine*>(idx.get())) {
            auto* flat = dynamic_cast<faiss::IndexFlat*>(rf->refine_index);
            if (!flat && !dynamic_cast<faiss::IndexScalarQuantizer*>(rf->refine_index)) {
                throw std::runtime_error("refine index is neither flat nor a scalar quantizer");
            }
            if (flat && nb_refine) {  // (a quantised refine index is dropped: *nb_refine stays 0)
                *nb_refine = flat->ntotal;
            }
            if (flat && xb_refine) {
                std::memcpy(xb_refine, flat->get_xb(), sizeof(float) * flat->ntotal * flat->d);
            }
            faiss::Index* base = rf->base_index;
            rf->base_index = nullptr;  // keep the base, drop wrapper + refine index
            delete rf->refine_index;
            rf->refine_index = nullptr;
            rf->own_fields = false;
            idx.reset(base);
        }
        h->d = idx->d;
        h->metric = idx->metric_type == faiss::METRIC_L2 ? 0 : 1;
        if (dynamic_cast<faiss::IndexIVFPQ*>(idx.get())) {
            h->kind = K_IVF_PQ;
        } else if (dynamic_cast<faiss::IndexIVFScalarQuantizer*>(idx.get())) {
            h->kind = K_IVF_SQ8;
        } else if (dynamic_cast<faiss::IndexIVFFlat*>(idx.get())) {
            h->kind = K_IVF_FLAT;
        } else if (dynamic_cast<faiss::IndexFlat*>(idx.get())) {
            h->kind = K_FLAT;
        } else {
            throw std::runtime_error("deserialized index kind not on the path");
        }
        h->index = std::move(idx);  // an IVF index read from bytes owns its quantizer (own_fields)
    });
    if (rc != 0) {
        delete h;
        return nullptr;
    }
    return h;
}

/// Range search driven the way IvfIndexNode::RangeSearch / FlatIndexNode::RangeSearch do (reference
/// src/index/ivf/ivf.cc:1231-1420): one query per faiss::RangeSearchResult(1),
/// IVFSearchParameters{nprobe = nlist, max_codes = 0, max_empty_result_buckets, sel}.
/// lims[nq + 1]; *out_ids / *out_dis are malloc'ed (release with ref_free).
int ref_range_search(
        void* hv,
        int64_t nq,
        const float* q,
        float radius,
        int64_t max_empty_result_buckets,
        const uint8_t* bitset,
        int64_t nbits,
        int64_t* lims,
        int64_t** out_ids,
        float** out_dis) {
    auto* h = static_cast<RefIndex*>(hv);
    return guarded([&] {
        std::vector<int64_t> ids;
        std::vector<float> dis;
        lims[0] = 0;
        for (int64_t i = 0; i < nq; i++) {
            faiss::RangeSearchResult res(1);
            std::unique_ptr<BitsetSelector> sel;
            if (bitset) {
                sel.reset(new BitsetSelector(bitset, nbits));
            }
            if (h->kind == K_FLAT) {
                faiss::SearchParameters sp;
                sp.sel = sel.get();
                h->index->range_search(1, q + i * h->d, radius, &res, &sp);
            } else {
                faiss::IVFSearchParameters sp;
                sp.nprobe = h->ivf()->nlist;
                sp.max_codes = 0;
                sp.max_empty_result_buckets = (size_t)max_empty_result_buckets;
                sp.sel = sel.get();
                h->index->range_search(1, q + i * h->d, radius, &res, &sp);
            }
            const size_t n = res.lims[1];
            ids.insert(ids.end(), res.labels, res.labels + n);
            dis.insert(dis.end(), res.distances, res.distances + n);
            lims[i + 1] = (int64_t)ids.size();
        }
        *out_ids = (int64_t*)malloc(sizeof(int64_t) * (ids.size() + 1));
        *out_dis = (float*)malloc(sizeof(float) * (dis.size() + 1));
        std::memcpy(*out_ids, ids.data(), sizeof(int64_t) * ids.size());
        std::memcpy(*out_dis, dis.data(), sizeof(float) * dis.size());
    });
}

void ref_free(void* p) {
    free(p);
}

/// coarse quantizer alone: quantizer->search(1, q, nprobe) per query
/// (reference thirdparty/faiss/faiss/IndexIVF.cpp:336-342).
int ref_coarse(void* hv, int64_t nq, const float* q, int64_t nprobe, float* D, int64_t* I) {
    auto* h = static_cast<RefIndex*>(hv);
    return guarded([&] {
        for (int64_t i = 0; i < nq; i++) {
            h->quantizer->search(1, q + i * h->d, nprobe, D + i * nprobe, I + i * nprobe);
        }
    });
}

/* ---------------- primitives (baseline faiss scalar build) ---------------- */

float ref_fvec_L2sqr(const float* x, const float* y, int64_t d) {
    return faiss::fvec_L2sqr(x, y, d);
}
float ref_fvec_inner_product(const float* x, const float* y, int64_t d) {
    return faiss::fvec_inner_product(x, y, d);
}
float ref_fvec_norm_L2sqr(const float* x, int64_t d) {
    return faiss::fvec_norm_L2sqr(x, d);
}
void ref_fvec_madd(int64_t n, const float* a, float bf, const float* b, float* c) {
    faiss::fvec_madd(n, a, bf, b, c);
}

}  // extern "C"
