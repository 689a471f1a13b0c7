// oracle/kref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// C-ABI over the index classes Knowhere's nodes actually instantiate for COSINE: the reference's FAISS fork
// (thirdparty/faiss/faiss/cppcontrib/knowhere: IndexFlatCosine, IndexIVFFlatCosine, its norm-bearing inverted lists,
// index_io) with src/common/utils.cc (NormalizeVecs / CopyAndNormalizeVecs) and the scalar SIMD hook table
// (ref_hooks.cpp -> src/simd/distances_ref.cc), all compiled where they lie into oracle/_ref/libknowhere_kref.so by
// oracle/Makefile (target kref).  The driver does what the nodes do and nothing else:
//   FlatIndexNode::Search   src/index/flat/flat.cc:98-122   query copied + normalised, IndexFlatCosine::search(1, ..)
//   IvfIndexNode::Train/Add src/index/ivf/ivf.cc:585-606, 820-850  IndexIVFFlatCosine::train / add_with_ids
//   IvfIndexNode::Search    src/index/ivf/ivf.cc:940-960    query copied + normalised, search(1, ..) with nprobe
// plus NormalizeDataset (ivf.cc:556-565) for the PQ / SQ kinds, and write_index so the node's Deserialize can be tested
// on bytes the reference wrote.
#include <faiss/cppcontrib/knowhere/IndexCosine.h>
#include <faiss/cppcontrib/knowhere/IndexFlat.h>
#include <faiss/cppcontrib/knowhere/IndexIVFFlat.h>
#include <faiss/cppcontrib/knowhere/index_io.h>
#include <faiss/cppcontrib/knowhere/invlists/InvertedLists.h>
#include <faiss/impl/io.h>

#include <omp.h>

#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "knowhere/bitsetview.h"
#include "knowhere/bitsetview_idselector.h"
#include "knowhere/utils.h"

namespace K = faiss::cppcontrib::knowhere;

namespace {
thread_local std::string g_err;
template <class Fn>
int guarded(Fn&& f) {
    try {
        f();
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}
struct KIvf {
    std::unique_ptr<K::IndexIVFFlatCosine> index;  // owns its quantizer (own_fields)
};
}  // namespace

extern "C" {

const char* kref_last_error() { return g_err.c_str(); }

// knowhere::NormalizeVecs: rows normalised in place, their norms returned (src/common/utils.cc:60-93)
int kref_normalize(float* x, int64_t n, int d, float* norms) {
    return guarded([&] {
        auto v = knowhere::NormalizeVecs<float>(x, (size_t)n, d);
        if (norms) std::memcpy(norms, v.data(), sizeof(float) * n);
    });
}

// FLAT + COSINE as FlatIndexNode: IndexFlatCosine over the RAW rows; one task per query
int kref_flat_cosine_search(int d, int64_t nb, const float* xb, int64_t nq, const float* xq, int64_t k,
                            const uint8_t* bitset, int64_t nbits, float* D, int64_t* I, float* inv_norms_out) {
    return guarded([&] {
        K::IndexFlatCosine index(d);
        index.add(nb, xb);
        if (inv_norms_out) std::memcpy(inv_norms_out, index.get_inverse_l2_norms(), sizeof(float) * nb);
        knowhere::BitsetView bv(bitset, (size_t)nbits);
        omp_set_num_threads(1);
        for (int64_t i = 0; i < nq; i++) {
            auto q = knowhere::CopyAndNormalizeVecs(xq + i * d, 1, d);
            knowhere::BitsetViewIDSelector sel(bv);
            faiss::SearchParameters sp;
            sp.sel = bitset ? &sel : nullptr;
            index.search(1, q.get(), k, D + i * k, I + i * k, &sp);
        }
    });
}

// IVF_FLAT + COSINE as IvfIndexNode<.., IndexIVFFlat>
void* kref_ivfflat_cosine_create(int d, int64_t nlist) {
    auto h = new KIvf();
    int rc = guarded([&] {
        auto qzr = std::make_unique<K::IndexFlat>(d, faiss::METRIC_INNER_PRODUCT);
        h->index = std::make_unique<K::IndexIVFFlatCosine>(qzr.get(), d, nlist, faiss::METRIC_INNER_PRODUCT);
        h->index->quantizer = qzr.release();
        h->index->own_fields = true;
    });
    if (rc) {
        delete h;
        return nullptr;
    }
    return h;
}
void kref_ivfflat_cosine_destroy(void* h) { delete static_cast<KIvf*>(h); }

int kref_ivfflat_cosine_train(void* hv, int64_t n, const float* x, int niter, int seed) {
    auto* h = static_cast<KIvf*>(hv);
    return guarded([&] {
        if (niter > 0) h->index->cp.niter = niter;
        if (seed >= 0) h->index->cp.seed = seed;
        h->index->train(n, x);
    });
}
int kref_ivfflat_cosine_get_centroids(void* hv, float* out) {
    auto* h = static_cast<KIvf*>(hv);
    return guarded([&] { h->index->quantizer->reconstruct_n(0, h->index->nlist, out); });
}
int kref_ivfflat_cosine_add(void* hv, int64_t n, const float* x) {
    auto* h = static_cast<KIvf*>(hv);
    return guarded([&] { h->index->add(n, x); });
}
int64_t kref_ivfflat_cosine_list_size(void* hv, int64_t l) {
    return (int64_t) static_cast<KIvf*>(hv)->index->invlists->list_size(l);
}
int kref_ivfflat_cosine_get_list(void* hv, int64_t l, uint8_t* codes, int64_t* ids, float* norms) {
    auto* h = static_cast<KIvf*>(hv);
    return guarded([&] {
        auto* il = h->index->invlists;
        const size_t n = il->list_size(l);
        if (!n) return;
        faiss::InvertedLists::ScopedCodes sc(il, l);
        faiss::InvertedLists::ScopedIds si(il, l);
        std::memcpy(codes, sc.get(), n * il->code_size);
        std::memcpy(ids, si.get(), n * sizeof(int64_t));
        auto* nil = dynamic_cast<const K::NormInvertedLists*>(il);
        FAISS_THROW_IF_NOT_MSG(nil, "cosine index without norm-bearing inverted lists");
        const float* p = nil->get_code_norms(l, 0);
        std::memcpy(norms, p, n * sizeof(float));
        nil->release_code_norms(l, p);
    });
}
int kref_ivfflat_cosine_search(void* hv, int64_t nq, const float* xq, int64_t k, int64_t nprobe, const uint8_t* bitset,
                               int64_t nbits, float* D, int64_t* I) {
    auto* h = static_cast<KIvf*>(hv);
    return guarded([&] {
        const int d = h->index->d;
        knowhere::BitsetView bv(bitset, (size_t)nbits);
        omp_set_num_threads(1);
        for (int64_t i = 0; i < nq; i++) {
            auto q = knowhere::CopyAndNormalizeVecs(xq + i * d, 1, d);
            knowhere::BitsetViewIDSelector sel(bv);
            faiss::IVFSearchParameters sp;
            sp.nprobe = nprobe;
            sp.max_codes = 0;
            sp.sel = bitset ? &sel : nullptr;
            h->index->search(1, q.get(), k, D + i * k, I + i * k, &sp);
        }
    });
}
// the bytes IvfIndexNode::Serialize writes (faiss fork write_index into a memory writer)
int64_t kref_ivfflat_cosine_serialize(void* hv, uint8_t* out, int64_t cap) {
    auto* h = static_cast<KIvf*>(hv);
    int64_t n = -1;
    guarded([&] {
        faiss::VectorIOWriter w;
        K::write_index(h->index.get(), &w);
        n = (int64_t)w.data.size();
        if (n <= cap) std::memcpy(out, w.data.data(), (size_t)n);
    });
    return n;
}
int64_t kref_flat_cosine_serialize(int d, int64_t nb, const float* xb, uint8_t* out, int64_t cap) {
    int64_t n = -1;
    guarded([&] {
        K::IndexFlatCosine index(d);
        index.add(nb, xb);
        faiss::VectorIOWriter w;
        K::write_index(&index, &w);
        n = (int64_t)w.data.size();
        if (n <= cap) std::memcpy(out, w.data.data(), (size_t)n);
    });
    return n;
}

}  // extern "C"
