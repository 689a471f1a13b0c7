// knowhere_amd/csrc/knhip_api_build.hip -- Train / Add on the device behind the C ABI (include/knhip.h "GPU build"): faiss
// Clustering restated over the kernels of build.hip, encoder training, assignment + encoding + append.
#include "knhip_internal.h"

namespace knhip_host {

// faiss RandomGenerator / rand_perm (thirdparty/faiss/faiss/utils/random.cpp:35-55, 188-199): std::mt19937 seeded with
// (unsigned)seed, rand_int(max) = mt() % max, Fisher-Yates from the front
void faiss_rand_perm(std::vector<int64_t>& perm, int64_t n, int64_t seed) {
    perm.resize((size_t)n);
    std::iota(perm.begin(), perm.end(), (int64_t)0);
    std::mt19937 mt((unsigned int)seed);
    for (int64_t i = 0; i + 1 < n; i++) {
        const int64_t i2 = i + (int64_t)(mt() % (uint32_t)(n - i));
        std::swap(perm[(size_t)i], perm[(size_t)i2]);
    }
}

// detail::split_clusters (impl/ClusteringHelpers.cpp:177-240), on the host copy of the centroids
int split_clusters_host(int d, int64_t k, int64_t n, std::vector<float>& hassign, std::vector<float>& cen) {
    constexpr float EPS = 1.f / 1024.f;
    int nsplit = 0;
    std::mt19937 mt(1234u);
    auto rand_float = [&]() { return mt() / float(mt.max()); };
    for (int64_t ci = 0; ci < k; ci++) {
        if (hassign[ci] != 0) {
            continue;
        }
        int64_t cj = 0;
        const int64_t max_tries = 10 * k;
        int64_t n_tries = 0;
        bool found = false;
        for (cj = 0; n_tries < max_tries; cj = (cj + 1) % k) {
            const float p = (float)((hassign[cj] - 1.0) / (float)(n - k));
            const float r = rand_float();
            if (r < p) {
                found = true;
                break;
            }
            n_tries++;
        }
        if (!found) {
            cj = 0;
            for (int64_t j = 1; j < k; j++) {
                if (hassign[j] > hassign[cj]) {
                    cj = j;
                }
            }
        }
        std::memcpy(&cen[(size_t)ci * d], &cen[(size_t)cj * d], sizeof(float) * d);
        for (int j = 0; j < d; j++) {
            if (j % 2 == 0) {
                cen[(size_t)ci * d + j] *= 1 + EPS;
                cen[(size_t)cj * d + j] *= 1 - EPS;
            } else {
                cen[(size_t)ci * d + j] *= 1 - EPS;
                cen[(size_t)cj * d + j] *= 1 + EPS;
            }
        }
        hassign[ci] = hassign[cj] / 2;
        hassign[cj] -= hassign[ci];
        nsplit++;
    }
    return nsplit;
}

// IndexFlat::assign (k = 1 search, first best centroid wins) of n device rows.  L2: the coarse search's canonical order
// (distance, then id ascending) is the reference's answer.  Inner product: canonical ties come highest id first, the
// reference keeps the lowest -- rows whose two best candidates tie are rescanned (build.hip).
int assign_rows(const knhip_index* idx, const float* d_x, int64_t n, int64_t* d_assign, hipStream_t s) {
    if (n <= 0) {
        return KNHIP_OK;
    }
    DevBuf dist;
    if (idx->desc.metric != KNHIP_IP || idx->nlist < 2) {
        HIP_TRY(dist.alloc((size_t)n * sizeof(float)));
        if (int rc = knhip_coarse_search_device(idx, d_x, n, 1, d_assign, dist.as<float>(), s)) return rc;
        HIP_TRY(hipStreamSynchronize(s)); // (the scratch is freed on return)
        return KNHIP_OK;
    }
    DevBuf keys2;
    HIP_TRY(dist.alloc((size_t)n * 2 * sizeof(float)));
    HIP_TRY(keys2.alloc((size_t)n * 2 * sizeof(int64_t)));
    if (int rc = knhip_coarse_search_device(idx, d_x, n, 2, keys2.as<int64_t>(), dist.as<float>(), s)) return rc;
    HIP_TRY(launch_assign_first_max_ip(d_x, n, idx->d, idx->centroids.as<float>(), idx->nlist, keys2.as<int64_t>(),
                                       dist.as<float>(), d_assign, s));
    HIP_TRY(hipStreamSynchronize(s)); // (the scratch above is freed on return)
    return KNHIP_OK;
}

struct TrainParams {
    int niter = 25;          // ClusteringParameters default (Clustering.h); the level-1 quantizer overrides it with 10
    int max_points = 256;
    int64_t seed = 1234;
    bool spherical = false;  // Clustering::post_process_centroids renormalises the centroids
};

// default_niter: 25 for a plain Clustering (PQ codebooks, knhip_kmeans_device), 10 for the level-1 quantizer of an IVF
// index (Level1Quantizer's constructor, IndexIVF.cpp:44).  default_spherical: what the caller's metric implies
// (IndexIVF's constructor switches it on for the inner product, IndexIVF.cpp:178-181).
TrainParams resolve(const knhip_train_params* p, int default_niter, bool default_spherical) {
    TrainParams t;
    t.niter = default_niter;
    t.spherical = default_spherical;
    if (p) {
        if (p->niter > 0) t.niter = p->niter;
        if (p->max_points_per_centroid > 0) t.max_points = p->max_points_per_centroid;
        if (p->seed != 0) t.seed = p->seed;
        if (p->spherical == 1) t.spherical = true;
        if (p->spherical == 2) t.spherical = false;
    }
    return t;
}

// Lloyd k-means as faiss Clustering::train_encoded runs it (Clustering.cpp:95-380, nredo = 1, RANDOM init, no weights):
// subsample to k * max_points rows (rand_perm(seed)), initial centroids = rows perm'[0..k) of rand_perm(seed + 1), then
// niter x { exact k = 1 search of every row, compute_centroids, split_clusters }.
// x: n rows of leading dimension ld, the clustered sub-vector = columns [off, off + d).  k <= 1024 and d <= 64 run on
// the LDS codebook kernel (L2 only: PQ sub-quantizers), everything else on a temporary coarse quantizer (needs ld == d).
int kmeans_impl(int device, int metric, int d, int64_t n, const float* d_x, int64_t ld, int off, int64_t k,
                const TrainParams& tp, float* d_cen) {
    if (n < k || k <= 0 || d <= 0) {
        return fail(KNHIP_ERR_INVALID_ARGS, "k-means needs at least as many training vectors as centroids");
    }
    // (the LDS codebook kernel keeps k x d floats in one workgroup's LDS: 160 KB on gfx950, 8 KB left for the compiler)
    const bool small = k <= 1024 && d <= 64 && metric == KNHIP_L2 && (size_t)k * d * sizeof(float) <= 152 * 1024;
    if (!small && (ld != d || off != 0)) {
        return fail(KNHIP_ERR_INVALID_ARGS, "k-means: strided input only with a small codebook");
    }
    DeviceGuard g(device);
    // -- training set: contiguous [nx][d]
    DevBuf xs_buf, rows_buf;
    const float* xs = d_x;
    int64_t nx = n;
    std::vector<int64_t> perm;
    const bool strided = (ld != d || off != 0);
    if (n > k * (int64_t)tp.max_points || strided) {
        if (n > k * (int64_t)tp.max_points) {
            faiss_rand_perm(perm, n, tp.seed);
            nx = k * (int64_t)tp.max_points;
            perm.resize((size_t)nx);
        } else {
            perm.resize((size_t)n);
            std::iota(perm.begin(), perm.end(), (int64_t)0);
        }
        if (strided) { // gather sub-vectors: address rows of length 1 float with explicit offsets
            std::vector<int64_t> idxs((size_t)nx * d);
            for (int64_t i = 0; i < nx; i++) {
                for (int j = 0; j < d; j++) {
                    idxs[(size_t)i * d + j] = perm[(size_t)i] * ld + off + j;
                }
            }
            if (int rc = upload(rows_buf, idxs.data(), idxs.size() * sizeof(int64_t))) return rc;
            HIP_TRY(xs_buf.alloc((size_t)nx * d * sizeof(float)));
            HIP_TRY(launch_gather_rows(d_x, rows_buf.as<int64_t>(), nx * d, 1, xs_buf.as<float>(), nullptr));
        } else {
            if (int rc = upload(rows_buf, perm.data(), perm.size() * sizeof(int64_t))) return rc;
            HIP_TRY(xs_buf.alloc((size_t)nx * d * sizeof(float)));
            HIP_TRY(launch_gather_rows(d_x, rows_buf.as<int64_t>(), nx, d, xs_buf.as<float>(), nullptr));
        }
        xs = xs_buf.as<float>();
    }
    if (nx == k) { // corner case of the reference: the training set IS the centroid table
        HIP_TRY(hipMemcpy(d_cen, xs, (size_t)k * d * sizeof(float), hipMemcpyDeviceToDevice));
        return KNHIP_OK;
    }
    // -- initial centroids
    {
        std::vector<int64_t> p2;
        faiss_rand_perm(p2, nx, tp.seed + 1);
        p2.resize((size_t)k);
        DevBuf pb;
        if (int rc = upload(pb, p2.data(), p2.size() * sizeof(int64_t))) return rc;
        HIP_TRY(launch_gather_rows(xs, pb.as<int64_t>(), k, d, d_cen, nullptr));
        HIP_TRY(hipDeviceSynchronize());
    }
    DevBuf inv_norm;
    if (tp.spherical) { // post_process_centroids after the initialisation (Clustering.cpp:251)
        HIP_TRY(inv_norm.alloc((size_t)k * sizeof(float)));
        HIP_TRY(launch_renorm_rows(d_cen, k, d, inv_norm.as<float>(), nullptr));
    }
    // -- iterations.  (The reference leaves the loop when the objective repeats bit for bit, Clustering.cpp:362-377 with
    // early_stop_threshold 0: that is the fixed point of this deterministic loop, where further iterations reproduce
    // the same centroids -- running them changes nothing.)
    DevBuf keys64, keys32, sorted_rows, seg_off, tmp, hassign_d;
    HIP_TRY(sorted_rows.alloc((size_t)nx * sizeof(int32_t)));
    HIP_TRY(seg_off.alloc((size_t)(k + 1) * sizeof(int64_t)));
    const size_t tmp_bytes = group_rows_tmp_bytes(nx, k);
    HIP_TRY(tmp.alloc(tmp_bytes));
    HIP_TRY(hassign_d.alloc((size_t)k * sizeof(float)));
    knhip_index* assigner = nullptr;
    struct Guard {
        knhip_index*& p;
        ~Guard() { knhip_index_destroy(p); }
    } guard{assigner};
    if (small) {
        HIP_TRY(keys32.alloc((size_t)nx * sizeof(int32_t)));
    } else {
        HIP_TRY(keys64.alloc((size_t)nx * sizeof(int64_t)));
        knhip_desc desc{};
        desc.kind = KNHIP_IVF_FLAT;
        desc.metric = metric;
        desc.dim = d;
        desc.device = device;
        desc.nlist = k;
        if (int rc = knhip_index_create(&desc, &assigner)) return rc;
    }
    std::vector<float> hassign((size_t)k), cen_h;
    for (int it = 0; it < tp.niter; it++) {
        if (small) {
            HIP_TRY(launch_nearest_small(xs, nx, d, 0, d, d_cen, (int)k, keys32.as<int32_t>(), nullptr));
            HIP_TRY(group_rows_by_key(nullptr, keys32.as<int32_t>(), nx, k, sorted_rows.as<int32_t>(),
                                      seg_off.as<int64_t>(), tmp.p, tmp_bytes, nullptr));
        } else {
            if (int rc = knhip_index_set_coarse_device(assigner, d_cen)) return rc;
            if (int rc = assign_rows(assigner, xs, nx, keys64.as<int64_t>(), nullptr)) return rc;
            HIP_TRY(group_rows_by_key(keys64.as<int64_t>(), nullptr, nx, k, sorted_rows.as<int32_t>(),
                                      seg_off.as<int64_t>(), tmp.p, tmp_bytes, nullptr));
        }
        HIP_TRY(launch_centroid_update(xs, d, 0, d, sorted_rows.as<int32_t>(), seg_off.as<int64_t>(), k, d_cen,
                                       hassign_d.as<float>(), nullptr));
        HIP_TRY(hipMemcpy(hassign.data(), hassign_d.p, (size_t)k * sizeof(float), hipMemcpyDeviceToHost));
        bool any_empty = false;
        for (int64_t c = 0; c < k; c++) {
            any_empty |= hassign[(size_t)c] == 0;
        }
        if (any_empty) {
            cen_h.resize((size_t)k * d);
            HIP_TRY(hipMemcpy(cen_h.data(), d_cen, cen_h.size() * sizeof(float), hipMemcpyDeviceToHost));
            split_clusters_host(d, k, nx, hassign, cen_h);
            HIP_TRY(hipMemcpy(d_cen, cen_h.data(), cen_h.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        if (tp.spherical) { // post_process_centroids after the update and the split (Clustering.cpp:347)
            HIP_TRY(launch_renorm_rows(d_cen, k, d, inv_norm.as<float>(), nullptr));
        }
    }
    HIP_TRY(hipDeviceSynchronize());
    return KNHIP_OK;
}

int set_pq_device(knhip_index* idx, const float* d_cb) {
    HIP_TRY(idx->cb.alloc((size_t)256 * idx->d * sizeof(float)));
    HIP_TRY(hipMemcpy(idx->cb.p, d_cb, (size_t)256 * idx->d * sizeof(float), hipMemcpyDeviceToDevice));
    idx->has_pq = true;
    idx->cb_t.release();
    if (pq_scan_q4_supports(idx->desc.pq_m, idx->d, 1)) {
        HIP_TRY(idx->cb_t.alloc((size_t)256 * idx->desc.pq_m * sizeof(float4)));
        HIP_TRY(launch_pq_cb_transpose(idx->cb.as<float>(), idx->desc.pq_m, idx->d / idx->desc.pq_m,
                                       idx->cb_t.as<float4>(), nullptr));
        HIP_TRY(hipDeviceSynchronize());
    }
    return maybe_build_precomp(idx);
}

// assignment (k = 1 exact coarse search) + codes of n device rows; d_assign [n] int64, d_codes [n][code_size]
// d_x_assign: rows the assignment is taken from when they differ from the stored ones (COSINE IVF-Flat: assigned by the
// normalised row, stored raw -- IndexIVFFlatCosine::add_with_ids, cppcontrib/knowhere/IndexIVFFlat.cpp:516-524)
int encode_rows(const knhip_index* idx, int64_t n, const float* d_x, int64_t* d_assign, uint8_t* d_codes, hipStream_t s,
                const float* d_x_assign = nullptr) {
    const int d = idx->d;
    DevBuf resid;
    if (int rc = assign_rows(idx, d_x_assign ? d_x_assign : d_x, n, d_assign, s)) return rc;
    if (idx->desc.kind == KNHIP_IVF_FLAT) {
        HIP_TRY(hipMemcpyAsync(d_codes, d_x, (size_t)n * d * sizeof(float), hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
        return KNHIP_OK;
    }
    // by_residual (IndexIVFPQ / IndexIVFScalarQuantizer default): encode x - centroid
    HIP_TRY(resid.alloc((size_t)std::max<int64_t>(n, 1) * d * sizeof(float)));
    HIP_TRY(launch_residual(d_x, idx->centroids.as<float>(), d_assign, n, d, resid.as<float>(), s));
    if (idx->desc.kind == KNHIP_IVF_PQ) {
        HIP_TRY(launch_pq_encode(resid.as<float>(), n, d, idx->desc.pq_m, idx->cb.as<float>(), d_codes, s));
    } else {
        HIP_TRY(launch_sq8_encode(resid.as<float>(), n, d, idx->sq_trained.as<float>(), d_codes, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    return KNHIP_OK;
}

int check_trained_for_add(const knhip_index* idx) {
    const int kind = idx->desc.kind;
    if (kind == KNHIP_BRUTE_FORCE) {
        return KNHIP_OK;
    }
    if (!idx->has_coarse) {
        return fail(KNHIP_ERR_NOT_TRAINED, "coarse centroids not set");
    }
    if (kind == KNHIP_IVF_PQ && !idx->has_pq) {
        return fail(KNHIP_ERR_NOT_TRAINED, "PQ codebooks not set");
    }
    if (kind == KNHIP_IVF_SQ8 && !idx->has_sq) {
        return fail(KNHIP_ERR_NOT_TRAINED, "SQ parameters not set");
    }
    return KNHIP_OK;
}

int add_device_impl(knhip_index* idx, int64_t n, const float* d_x, const int64_t* d_ids,
                    const float* d_x_assign = nullptr) {
    if (n == 0) {
        return KNHIP_OK;
    }
    const int d = idx->d;
    if (idx->desc.kind == KNHIP_BRUTE_FORCE) {
        if (d_ids) {
            return fail(KNHIP_ERR_NOT_IMPLEMENTED, "explicit ids on a brute-force index (ids are row + offset)");
        }
        // append to the raw rows, lay the whole base out again
        const int64_t n0 = idx->ntotal;
        DevBuf all;
        HIP_TRY(all.alloc((size_t)(n0 + n) * d * sizeof(float)));
        if (n0) {
            HIP_TRY(hipMemcpy(all.p, idx->codes_aos.p, (size_t)n0 * d * sizeof(float), hipMemcpyDeviceToDevice));
        }
        HIP_TRY(hipMemcpy(all.as<float>() + n0 * d, d_x, (size_t)n * d * sizeof(float), hipMemcpyDeviceToDevice));
        std::swap(idx->codes_aos.p, all.p);
        std::swap(idx->codes_aos.bytes, all.bytes);
        return add_vectors_common(idx, n0 + n, idx->codes_aos.as<float>(), nullptr, idx->id_offset);
    }
    if (int rc = check_trained_for_add(idx)) return rc;
    const int64_t nlist = idx->nlist;
    const int64_t cs = idx->code_size;
    const int64_t n0 = idx->has_data ? idx->ntotal : 0;
    DevBuf assign, codes, ids_new, sorted_rows, seg_off, tmp;
    HIP_TRY(assign.alloc((size_t)n * sizeof(int64_t)));
    HIP_TRY(codes.alloc((size_t)n * cs));
    if (int rc = encode_rows(idx, n, d_x, assign.as<int64_t>(), codes.as<uint8_t>(), nullptr, d_x_assign)) return rc;
    const int64_t* new_ids = d_ids;
    if (!d_ids) { // Knowhere ids are the running row numbers (IvfIndexNode::Add -> add_core without xids)
        HIP_TRY(ids_new.alloc((size_t)n * sizeof(int64_t)));
        HIP_TRY(launch_iota_i64(ids_new.as<int64_t>(), n, n0, nullptr));
        new_ids = ids_new.as<int64_t>();
    }
    HIP_TRY(sorted_rows.alloc((size_t)n * sizeof(int32_t)));
    HIP_TRY(seg_off.alloc((size_t)(nlist + 1) * sizeof(int64_t)));
    const size_t tmp_bytes = group_rows_tmp_bytes(n, nlist);
    HIP_TRY(tmp.alloc(tmp_bytes));
    HIP_TRY(group_rows_by_key(assign.as<int64_t>(), nullptr, n, nlist, sorted_rows.as<int32_t>(), seg_off.as<int64_t>(),
                              tmp.p, tmp_bytes, nullptr));
    std::vector<int64_t> seg((size_t)nlist + 1);
    HIP_TRY(hipMemcpy(seg.data(), seg_off.p, seg.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
    std::vector<int64_t> off((size_t)nlist + 1, 0);
    for (int64_t l = 0; l < nlist; l++) {
        const int64_t old_len = n0 ? idx->h_list_off[l + 1] - idx->h_list_off[l] : 0;
        off[l + 1] = off[l] + old_len + (seg[l + 1] - seg[l]);
    }
    DevBuf out_off, old_off, out_codes, out_ids;
    if (int rc = upload(out_off, off.data(), off.size() * sizeof(int64_t))) return rc;
    if (n0) {
        if (int rc = upload(old_off, idx->h_list_off.data(), idx->h_list_off.size() * sizeof(int64_t))) return rc;
    }
    if (n0) {
        if (int rc = ensure_aos(idx)) return rc;
    }
    HIP_TRY(out_codes.alloc((size_t)(n0 + n) * cs));
    HIP_TRY(out_ids.alloc((size_t)(n0 + n) * sizeof(int64_t)));
    HIP_TRY(launch_merge_lists(n0 ? idx->codes_aos.as<uint8_t>() : nullptr, n0 ? idx->ids.as<int64_t>() : nullptr,
                               n0 ? old_off.as<int64_t>() : nullptr, codes.as<uint8_t>(), new_ids,
                               sorted_rows.as<int32_t>(), seg_off.as<int64_t>(), out_off.as<int64_t>(), nlist, cs,
                               out_codes.as<uint8_t>(), out_ids.as<int64_t>(), nullptr));
    HIP_TRY(hipDeviceSynchronize());
    // hand the merged arrays to the index (no further copy) and lay the kernel layouts out again
    std::swap(idx->codes_aos.p, out_codes.p);
    std::swap(idx->codes_aos.bytes, out_codes.bytes);
    std::swap(idx->ids.p, out_ids.p);
    std::swap(idx->ids.bytes, out_ids.bytes);
    return build_list_layout(idx, off, idx->codes_aos.as<uint8_t>(), idx->ids.as<int64_t>());
}

int train_device_impl(knhip_index* idx, int64_t n, const float* d_x, const knhip_train_params* p) {
    const int kind = idx->desc.kind;
    if (kind == KNHIP_BRUTE_FORCE) {
        return KNHIP_OK;
    }
    if (n <= 0 || !d_x) {
        return fail(KNHIP_ERR_INVALID_ARGS, "train: no training vectors");
    }
    const TrainParams tp = resolve(p, 10, idx->desc.metric == KNHIP_IP);
    const int d = idx->d;
    const int64_t nlist = idx->nlist;
    const int dev = idx->desc.device;
    DeviceGuard g(dev);
    // 1. level-1 quantizer (Level1Quantizer::train_q1, IndexIVF.cpp:55-121): k-means assigned by the index's own metric
    if (!idx->has_coarse) {
        DevBuf cen;
        HIP_TRY(cen.alloc((size_t)nlist * d * sizeof(float)));
        if (int rc = kmeans_impl(dev, idx->desc.metric, d, n, d_x, d, 0, nlist, tp, cen.as<float>())) return rc;
        if (int rc = knhip_index_set_coarse_device(idx, cen.as<float>())) return rc;
    }
    if (kind == KNHIP_IVF_FLAT) {
        return KNHIP_OK;
    }
    // 2. encoder training set (IndexIVF::train_encoder, IndexIVF.cpp:1213-1270): at most train_encoder_num_vectors()
    //    rows (IVF_PQ: 256 * ksub = 65536, IndexIVFPQ.cpp:97-99; IVF_SQ: 100000, IndexScalarQuantizer.cpp:159-161),
    //    the first rows of rand_perm(n, 1234) (fvecs_maybe_subsample, utils/utils.cpp:464-489); residuals to the
    //    assigned centroid (by_residual)
    const int64_t max_nt = kind == KNHIP_IVF_PQ ? (int64_t)256 * idx->ksub : 100000;
    int64_t nt = n;
    DevBuf xt_buf;
    const float* xt = d_x;
    if (n > max_nt) {
        std::vector<int64_t> perm;
        faiss_rand_perm(perm, n, 1234);
        nt = max_nt;
        perm.resize((size_t)nt);
        DevBuf pb;
        if (int rc = upload(pb, perm.data(), perm.size() * sizeof(int64_t))) return rc;
        HIP_TRY(xt_buf.alloc((size_t)nt * d * sizeof(float)));
        HIP_TRY(launch_gather_rows(d_x, pb.as<int64_t>(), nt, d, xt_buf.as<float>(), nullptr));
        HIP_TRY(hipDeviceSynchronize());
        xt = xt_buf.as<float>();
    }
    DevBuf assign, resid;
    HIP_TRY(assign.alloc((size_t)nt * sizeof(int64_t)));
    HIP_TRY(resid.alloc((size_t)nt * d * sizeof(float)));
    if (int rc = assign_rows(idx, xt, nt, assign.as<int64_t>(), nullptr)) return rc;
    HIP_TRY(launch_residual(xt, idx->centroids.as<float>(), assign.as<int64_t>(), nt, d, resid.as<float>(), nullptr));
    HIP_TRY(hipDeviceSynchronize());
    if (kind == KNHIP_IVF_PQ) {
        // ProductQuantizer::train (impl/ProductQuantizer.cpp:130-215, Train_default): one k-means per sub-space on its
        // slice of the residuals, ksub = 2^nbits centroids, ClusteringParameters defaults (25 iterations, seed 1234), L2
        const int M = idx->desc.pq_m, dsub = d / M, ksub = idx->ksub;
        DevBuf cb;
        HIP_TRY(cb.alloc((size_t)256 * d * sizeof(float)));
        TrainParams pq_tp;  // (the coarse quantizer's parameters do not apply to the codebooks)
        for (int m = 0; m < M; m++) {
            float* cbm = cb.as<float>() + (size_t)m * 256 * dsub;
            if (int rc = kmeans_impl(dev, KNHIP_L2, dsub, nt, resid.as<float>(), d, m * dsub, ksub, pq_tp, cbm)) return rc;
            for (int c = ksub; c < 256; c++) { // (entries no code refers to: copies of entry 0, see pad_codebook)
                HIP_TRY(hipMemcpy(cbm + (size_t)c * dsub, cbm, (size_t)dsub * sizeof(float), hipMemcpyDeviceToDevice));
            }
        }
        return set_pq_device(idx, cb.as<float>());
    }
    // ScalarQuantizer::train, QT_8bit, RS_minmax with rangestat_arg = 0 (impl/ScalarQuantizer.cpp train_NonUniform):
    // vmin = column minimum, vdiff = column maximum - vmin
    DevBuf mm;
    HIP_TRY(mm.alloc((size_t)2 * d * sizeof(float)));
    HIP_TRY(launch_col_minmax(resid.as<float>(), nt, d, mm.as<float>(), mm.as<float>() + d, nullptr));
    std::vector<float> h((size_t)2 * d);
    HIP_TRY(hipMemcpy(h.data(), mm.p, h.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int j = 0; j < d; j++) {
        h[(size_t)d + j] = h[(size_t)d + j] - h[(size_t)j];
    }
    return knhip_index_set_sq(idx, h.data(), h.data() + d);
}

} // namespace knhip_host

extern "C" {

int knhip_kmeans_device(int32_t metric, int32_t dim, int64_t n, const float* d_x, int64_t k,
                        const knhip_train_params* params, float* d_centroids, int32_t device) {
    if (!d_x || !d_centroids || (metric != KNHIP_L2 && metric != KNHIP_IP)) {
        return fail(KNHIP_ERR_INVALID_ARGS, "kmeans: bad arguments");
    }
    return kmeans_impl(device, metric, dim, n, d_x, dim, 0, k, resolve(params, 25, false), d_centroids);
}

int knhip_index_train_device(knhip_index* idx, int64_t n, const float* d_x, const knhip_train_params* params) {
    if (int rc = check_index(idx)) return rc;
    return train_device_impl(idx, n, d_x, params);
}

int knhip_index_train(knhip_index* idx, int64_t n, const float* x, const knhip_train_params* params) {
    if (int rc = check_index(idx)) return rc;
    if (idx->desc.kind == KNHIP_BRUTE_FORCE) {
        return KNHIP_OK;
    }
    if (n <= 0 || !x) {
        return fail(KNHIP_ERR_INVALID_ARGS, "train: no training vectors");
    }
    DeviceGuard g(idx->desc.device);
    // Only the rows k-means and the encoder can use are uploaded: the reference subsamples to
    // max(nlist * max_points_per_centroid, train_encoder_num_vectors) rows of two rand_perm draws; both draws index
    // the SAME caller array, so upload the union once is not possible without changing the draws -> upload all rows
    // when they fit half of the free HBM, otherwise fail loudly (stream in slices via knhip_index_train_device).
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const size_t bytes = (size_t)n * idx->d * sizeof(float);
    if (bytes > free_b / 2) {
        return fail(KNHIP_ERR_OUT_OF_MEMORY, "train: training set does not fit the device; pass a subsample");
    }
    DevBuf dx;
    if (int rc = upload(dx, x, bytes)) return rc;
    return train_device_impl(idx, n, dx.as<float>(), params);
}

int knhip_index_add_device(knhip_index* idx, int64_t n, const float* d_x, const int64_t* d_ids) {
    if (int rc = check_index(idx)) return rc;
    if (n < 0 || (n > 0 && !d_x)) {
        return fail(KNHIP_ERR_INVALID_ARGS, "add: bad arguments");
    }
    DeviceGuard g(idx->desc.device);
    std::lock_guard<std::mutex> lk(idx->add_mu);
    return add_device_impl(idx, n, d_x, d_ids);
}

int knhip_index_add(knhip_index* idx, int64_t n, const float* x, const int64_t* ids) {
    if (int rc = check_index(idx)) return rc;
    if (n < 0 || (n > 0 && !x)) {
        return fail(KNHIP_ERR_INVALID_ARGS, "add: bad arguments");
    }
    DeviceGuard g(idx->desc.device);
    std::lock_guard<std::mutex> lk(idx->add_mu);
    // slices of at most 1 GiB of rows keep the staging buffer bounded
    const int64_t step = std::max<int64_t>(1, ((int64_t)1 << 30) / ((int64_t)idx->d * 4));
    if (idx->desc.kind == KNHIP_BRUTE_FORCE || n <= step) {
        DevBuf dx, di;
        if (int rc = upload(dx, x, (size_t)n * idx->d * sizeof(float))) return rc;
        if (ids) {
            if (int rc = upload(di, ids, (size_t)n * sizeof(int64_t))) return rc;
        }
        return add_device_impl(idx, n, dx.as<float>(), ids ? di.as<int64_t>() : nullptr);
    }
    for (int64_t i0 = 0; i0 < n; i0 += step) {
        const int64_t m = std::min(step, n - i0);
        DevBuf dx, di;
        if (int rc = upload(dx, x + i0 * idx->d, (size_t)m * idx->d * sizeof(float))) return rc;
        if (ids) {
            if (int rc = upload(di, ids + i0, (size_t)m * sizeof(int64_t))) return rc;
        }
        if (int rc = add_device_impl(idx, m, dx.as<float>(), ids ? di.as<int64_t>() : nullptr)) return rc;
    }
    return KNHIP_OK;
}

int knhip_index_add_assigned_by(knhip_index* idx, int64_t n, const float* x_store, const float* x_assign,
                                const int64_t* ids) {
    if (int rc = check_index(idx)) return rc;
    if (idx->desc.kind != KNHIP_IVF_FLAT || n < 0 || (n > 0 && (!x_store || !x_assign))) {
        return fail(KNHIP_ERR_INVALID_ARGS, "add_assigned_by: IVF_FLAT index, two row arrays");
    }
    if (n == 0) {
        return KNHIP_OK;
    }
    DeviceGuard g(idx->desc.device);
    std::lock_guard<std::mutex> lk(idx->add_mu);
    // slices of at most 1 GiB per row array keep the staging buffers bounded, as knhip_index_add does
    const int64_t step = std::max<int64_t>(1, ((int64_t)1 << 30) / ((int64_t)idx->d * 4));
    for (int64_t i0 = 0; i0 < n; i0 += step) {
        const int64_t m = std::min(step, n - i0);
        DevBuf dx, da, di;
        if (int rc = upload(dx, x_store + i0 * idx->d, (size_t)m * idx->d * sizeof(float))) return rc;
        if (int rc = upload(da, x_assign + i0 * idx->d, (size_t)m * idx->d * sizeof(float))) return rc;
        if (ids) {
            if (int rc = upload(di, ids + i0, (size_t)m * sizeof(int64_t))) return rc;
        }
        if (int rc = add_device_impl(idx, m, dx.as<float>(), ids ? di.as<int64_t>() : nullptr, da.as<float>())) return rc;
    }
    return KNHIP_OK;
}

int knhip_index_assign(const knhip_index* idx, int64_t n, const float* x, int64_t* assign) {
    if (int rc = check_index(idx)) return rc;
    if (idx->desc.kind == KNHIP_BRUTE_FORCE || n < 0 || (n > 0 && (!x || !assign))) {
        return fail(KNHIP_ERR_INVALID_ARGS, "assign: IVF index, rows and output required");
    }
    if (!idx->has_coarse) {
        return fail(KNHIP_ERR_NOT_TRAINED, "coarse centroids not set");
    }
    DeviceGuard g(idx->desc.device);
    const int64_t step = std::max<int64_t>(1, ((int64_t)1 << 30) / ((int64_t)idx->d * 4));
    for (int64_t i0 = 0; i0 < n; i0 += step) {
        const int64_t m = std::min(step, n - i0);
        DevBuf dx, da;
        if (int rc = upload(dx, x + i0 * idx->d, (size_t)m * idx->d * sizeof(float))) return rc;
        HIP_TRY(da.alloc((size_t)m * sizeof(int64_t)));
        if (int rc = assign_rows(idx, dx.as<float>(), m, da.as<int64_t>(), nullptr)) return rc;
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(assign + i0, da.p, (size_t)m * sizeof(int64_t), hipMemcpyDeviceToHost));
    }
    return KNHIP_OK;
}

int knhip_index_encode_device(const knhip_index* idx, int64_t n, const float* d_x, int64_t* d_assign, uint8_t* d_codes,
                              void* stream) {
    if (int rc = check_index(idx)) return rc;
    if (idx->desc.kind == KNHIP_BRUTE_FORCE || n < 0 || (n > 0 && (!d_x || !d_assign || !d_codes))) {
        return fail(KNHIP_ERR_INVALID_ARGS, "encode: bad arguments");
    }
    if (int rc = check_trained_for_add(idx)) return rc;
    if (n == 0) {
        return KNHIP_OK;
    }
    DeviceGuard g(idx->desc.device);
    return encode_rows(idx, n, d_x, d_assign, d_codes, static_cast<hipStream_t>(stream));
}

} // extern "C"
