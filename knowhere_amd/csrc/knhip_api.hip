// knowhere_amd/csrc/knhip_api.hip -- the C ABI of libknhip.so (include/knhip.h): index object,
// HBM layouts, and the Search() orchestration over the kernels in this directory.
//
// Search() on the device, IVF kinds (mirrors faiss::IndexIVF::search,
// reference thirdparty/faiss/faiss/IndexIVF.cpp:305-399, driven per batch instead of per query):
//   1. coarse   : exact query x centroid distances (flat_full) + per-row top-nprobe (row_select)
//                 == quantizer->search(n, x, nprobe)                    IndexIVF.cpp:336-342
//   2. group    : (query, probe) -> list-major work items                worktable.hip
//   3. tables   : PQ query tables <q_m, cb[m][c]>                        IVFPQ_QueryTables.cpp:56-67
//   4. scan     : per-list code scan with per-(query, probe) top-k      search_preassigned :625-671
//   5. merge    : per query, k best of its nprobe partial lists          heap_reorder :665
// BRUTE_FORCE is steps 4-5 with base chunks in place of lists.
// Nothing in this path touches the host between the first and the last kernel.
#include "knhip_internal.h"

namespace knhip_host {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

int build_coarse_layout(knhip_index* idx) {
    const int nchunk = (idx->d + 3) / 4;
    const int64_t nblk = (idx->nlist + 63) / 64;
    HIP_TRY(idx->centroids_il.alloc((size_t)nblk * nchunk * 64 * sizeof(float4)));
    HIP_TRY(launch_interleave_rows(idx->centroids.as<float>(), idx->nlist, idx->d,
                                   idx->centroids_il.as<float4>(), 0, nullptr));
    HIP_TRY(idx->cnorm.alloc((size_t)idx->nlist * sizeof(float)));
    HIP_TRY(launch_row_norms(idx->centroids.as<float>(), idx->nlist, idx->d, idx->cnorm.as<float>(), nullptr));
    // the centroids as split bf16 operand rows of the coarse prefilter (coarse_gemm.hip)
    HIP_TRY(idx->centroids_bs.alloc((size_t)idx->nlist * coarse_bf16_slabs(idx->d) * 128));
    HIP_TRY(launch_coarse_bf16_split(idx->centroids.as<float>(), idx->nlist, idx->d, idx->centroids_bs.p, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    std::vector<float> cn((size_t)idx->nlist);
    HIP_TRY(hipMemcpy(cn.data(), idx->cnorm.p, cn.size() * sizeof(float), hipMemcpyDeviceToHost));
    idx->cnorm_max = cn.empty() ? 0.f : *std::max_element(cn.begin(), cn.end());
    idx->coarse_gemm = env_layout().coarse_gemm;
    return KNHIP_OK;
}

// Coarse quantizer for one batch: keys/cdis [nq][nprobe], best-first, bit-equal to the exact search.
// the rows a "nearest rows of every query" stage runs over: the coarse quantizer's centroids, or a chunk of a BRUTE_FORCE base
struct CoarseRows {
    const float* rows;      // [n][d] row major
    const float4* rows_il;  // interleaved 64-row blocks
    const void* rows_bs;    // split bf16 operand rows (null: the bf16 prefilter is not available)
    const float* norm;      // [n] ||x||^2
    float norm_max;
    int64_t n;
};

// a search over several row sets one after the other (the chunks of a BRUTE_FORCE base): the k-th best distance found so far
// bounds what a later chunk can contribute, so only the FIRST chunk needs the two-pass form (group minima -> bound ->
// candidates); the others take the running k-th, widened by the prefilter's eps, as their selection bound and make ONE pass
struct RowsRun {
    float* kth;          // [nq] running k-th best exact distance (worst value before the first chunk)
    bool have_bound;     // a chunk has been searched: kth bounds this one
    bool queries_ready;  // the queries' norms and split operand rows of this batch are in the workspace already
};

int coarse_rows_stage(const knhip_index* idx, Workspace* ws, const CoarseRows& R, const float* d_q, int64_t nq, int nprobe,
                      int64_t* keys, float* cdis, hipStream_t s, RowsRun* run = nullptr) {
    const int64_t nlist = R.n;
    const int d = idx->d;
    const bool is_l2 = idx->is_l2;
    HIP_TRY(ws->coarse_full.reserve((size_t)nq * nlist * sizeof(float)));
    FlatScanArgs c{};
    c.rows = R.rows_il;
    c.nrows = nlist;
    c.chunk_rows = 1024;
    c.d = d;
    c.nchunk = (d + 3) / 4;
    c.queries = d_q;
    c.nq = nq;
    const int margin = std::max(32, nprobe / 4);
    const int64_t ncand = (int64_t)nprobe + margin;
    // (the MFMA prefilter + exact re-rank is validated for up to 4096 candidates per query; above, the exact kernel)
    if (!idx->coarse_gemm || ncand >= nlist || ncand > 4096) {
        HIP_TRY(launch_flat_full(c, is_l2, ws->coarse_full.as<float>(), nullptr, 0, nullptr, s));
        unsigned long long* sort_scratch = nullptr;
        if ((size_t)nprobe > row_select_lds_max_k()) { // (more keys than the LDS sorts: a scratch row per query)
            int64_t kp = 2;
            while (kp < nprobe) {
                kp <<= 1;
            }
            HIP_TRY(ws->rs_sort.reserve((size_t)nq * kp * sizeof(unsigned long long)));
            sort_scratch = ws->rs_sort.as<unsigned long long>();
        }
        HIP_TRY(launch_row_select(ws->coarse_full.as<float>(), nq, nlist, nprobe, is_l2, keys, cdis, nullptr, s,
                                  sort_scratch));
        return KNHIP_OK;
    }
    HIP_TRY(ws->qnorm.reserve((size_t)nq * sizeof(float)));
    HIP_TRY(ws->cand_keys.reserve((size_t)nq * ncand * sizeof(int64_t)));
    HIP_TRY(ws->cand_approx.reserve((size_t)nq * ncand * sizeof(float)));
    HIP_TRY(ws->fail_flags.reserve(((size_t)nq + 1) * sizeof(int32_t))); // (+ the "any flag" summary)
    const bool q_ready = run != nullptr && run->queries_ready;
    if (!q_ready) {
        HIP_TRY(launch_row_norms(d_q, nq, d, ws->qnorm.as<float>(), s));
    }
    if (idx->coarse_gemm == 2 && R.rows_bs != nullptr && coarse_bf16_supports(nlist, (int)ncand)) {
        // bf16 matrix pipe, selection fused, no nq x nlist matrix (coarse_gemm.hip, round 5)
        int cap = 256;
        while (cap < 2 * ncand) {
            cap <<= 1;
        }
        HIP_TRY(ws->cg_gmin.reserve((size_t)coarse_bf16_groups(nlist) * nq * sizeof(float)));
        HIP_TRY(ws->cg_bound.reserve((size_t)nq * sizeof(float)));
        HIP_TRY(ws->cg_cnt.reserve((size_t)nq * sizeof(int32_t)));
        HIP_TRY(ws->cand_keys.reserve((size_t)nq * cap * sizeof(int64_t)));
        HIP_TRY(ws->cg_qs.reserve((size_t)nq * coarse_bf16_slabs(d) * 128));
        if (!q_ready) {
            HIP_TRY(launch_coarse_bf16_split(d_q, nq, d, ws->cg_qs.p, s));
        }
        const bool bound_given = run != nullptr && run->have_bound;
        if (bound_given) {
            HIP_TRY(launch_coarse_ext_bound(run->kth, nullptr, nprobe, nq, ws->qnorm.as<float>(), R.norm_max, d, is_l2,
                                            ws->cg_bound.as<float>(), s));
        }
        HIP_TRY(launch_coarse_bf16(ws->cg_qs.p, ws->qnorm.as<float>(), R.rows_bs, R.norm, nq, nlist,
                                   d, is_l2, (int)ncand, cap, ws->cg_gmin.as<float>(), ws->cg_bound.as<float>(),
                                   ws->cg_cnt.as<int32_t>(), ws->cand_keys.as<int64_t>(), s, bound_given));
        HIP_TRY(launch_coarse_rerank(d_q, R.rows, d, nq, nlist, cap, ws->cand_keys.as<int64_t>(), nullptr,
                                     nprobe, is_l2, ws->qnorm.as<float>(), R.norm_max, keys, cdis,
                                     ws->fail_flags.as<int32_t>(), idx->coarse_fail_dev.as<unsigned long long>(), s,
                                     ws->cg_cnt.as<int32_t>(), ws->cg_bound.as<float>(), /*need_kth=*/!bound_given));
        // exact fallback, restricted on the device to the flagged queries (normally none)
        HIP_TRY(launch_flat_full(c, is_l2, ws->coarse_full.as<float>(), nullptr, 0, ws->fail_flags.as<int32_t>(), s));
        HIP_TRY(launch_row_select(ws->coarse_full.as<float>(), nq, nlist, nprobe, is_l2, keys, cdis,
                                  ws->fail_flags.as<int32_t>(), s));
        if (run != nullptr) { // this chunk's k-th best (where it found k rows) tightens the running one
            HIP_TRY(launch_coarse_ext_bound(run->kth, cdis, nprobe, nq, ws->qnorm.as<float>(), R.norm_max, d, is_l2,
                                            ws->cg_bound.as<float>(), s));
            run->have_bound = true;
            run->queries_ready = true;
        }
        return KNHIP_OK;
    }
    if (run != nullptr) {
        run->have_bound = false; // (the other forms of the stage keep no running bound)
        run->queries_ready = false;
    }
    HIP_TRY(launch_coarse_gemm(d_q, ws->qnorm.as<float>(), R.rows, R.norm, nq,
                               nlist, d, is_l2, ws->coarse_full.as<float>(), s));
    if (row_select_thr_supports(nlist, (int)ncand)) {
        // two passes over the row (group minima -> bound -> candidates); rows with masses of equal values fall to the radix
        // select through the flags
        HIP_TRY(ws->rs_ovf.reserve((size_t)nq * sizeof(int32_t)));
        HIP_TRY(launch_row_select_thr(ws->coarse_full.as<float>(), nq, nlist, (int)ncand, is_l2,
                                      ws->cand_keys.as<int64_t>(), ws->cand_approx.as<float>(), ws->rs_ovf.as<int32_t>(), s));
        HIP_TRY(launch_row_select(ws->coarse_full.as<float>(), nq, nlist, (int)ncand, is_l2,
                                  ws->cand_keys.as<int64_t>(), ws->cand_approx.as<float>(), ws->rs_ovf.as<int32_t>(), s));
    } else {
        HIP_TRY(launch_row_select(ws->coarse_full.as<float>(), nq, nlist, (int)ncand, is_l2,
                                  ws->cand_keys.as<int64_t>(), ws->cand_approx.as<float>(), nullptr, s));
    }
    HIP_TRY(launch_coarse_rerank(d_q, R.rows, d, nq, nlist, (int)ncand,
                                 ws->cand_keys.as<int64_t>(), ws->cand_approx.as<float>(), nprobe, is_l2,
                                 ws->qnorm.as<float>(), R.norm_max, keys, cdis, ws->fail_flags.as<int32_t>(),
                                 idx->coarse_fail_dev.as<unsigned long long>(), s));
    // exact fallback, restricted on the device to the flagged queries (normally none)
    HIP_TRY(launch_flat_full(c, is_l2, ws->coarse_full.as<float>(), nullptr, 0, ws->fail_flags.as<int32_t>(), s));
    HIP_TRY(launch_row_select(ws->coarse_full.as<float>(), nq, nlist, nprobe, is_l2, keys, cdis,
                              ws->fail_flags.as<int32_t>(), s));
    return KNHIP_OK;
}

int coarse_stage(const knhip_index* idx, Workspace* ws, const float* d_q, int64_t nq, int nprobe,
                 int64_t* keys, float* cdis, hipStream_t s) {
    CoarseRows R{idx->centroids.as<float>(), idx->centroids_il.as<float4>(), idx->centroids_bs.p, idx->cnorm.as<float>(),
                 idx->cnorm_max, idx->nlist};
    return coarse_rows_stage(idx, ws, R, d_q, nq, nprobe, keys, cdis, s);
}

int maybe_build_precomp(knhip_index* idx) {
    // reference thirdparty/faiss/faiss/IndexIVFPQ.cpp:428-456: L2 + by_residual, table within limit
    idx->use_precomp = 0;
    idx->precomp_t.release();
    if (idx->desc.kind != KNHIP_IVF_PQ || !idx->has_coarse || !idx->has_pq || !idx->is_l2) {
        return KNHIP_OK;
    }
    const size_t limit = idx->desc.precomputed_table_max_bytes > 0
            ? (size_t)idx->desc.precomputed_table_max_bytes
            : ((size_t)1 << 31);
    const size_t table = (size_t)idx->nlist * 256 * idx->desc.pq_m * sizeof(float);
    // (the decision follows the REFERENCE's table, M x ksub x nlist floats: it selects the arithmetic of the distances)
    if ((size_t)idx->nlist * idx->ksub * idx->desc.pq_m * sizeof(float) > limit) {
        return KNHIP_OK;
    }
    HIP_TRY(idx->precomp_t.alloc(table));
    HIP_TRY(launch_pq_precomp_table(idx->centroids.as<float>(), idx->cb.as<float>(), idx->d,
                                    idx->desc.pq_m, idx->nlist, idx->precomp_t.as<float>(), nullptr));
    HIP_TRY(hipDeviceSynchronize());
    idx->use_precomp = 1;
    return KNHIP_OK;
}

// IVF_PQ: the skewed code layout of the systolic kernel (pq_scan.hip), from the canonical AoS codes
int build_pq_skew(const knhip_index* cidx) {
    knhip_index* idx = const_cast<knhip_index*>(cidx);
    std::lock_guard<std::mutex> lk(idx->mu);
    if (idx->skew_ready) {
        return KNHIP_OK;
    }
    const int M = idx->desc.pq_m;
    const int64_t nlist = idx->nlist;
    std::vector<int64_t> blk_off(nlist + 1, 0);
    for (int64_t l = 0; l < nlist; l++) {
        blk_off[l + 1] = blk_off[l] + pq_skew_blocks(idx->h_list_len[l], M);
    }
    HIP_TRY(idx->rows.alloc((size_t)blk_off[nlist] * M * sizeof(uint4)));
    HIP_TRY(launch_pq_skew_codes(idx->codes_aos.as<uint8_t>(), idx->d_list_row_off.as<int64_t>(),
                                 idx->d_list_len.as<int64_t>(), idx->d_list_blk_off.as<int64_t>(), nlist, M,
                                 idx->rows.as<uint4>(), nullptr));
    HIP_TRY(hipDeviceSynchronize());
    idx->skew_ready = true;
    return KNHIP_OK;
}

// flat / SQ8 indexes above this many code bytes keep only the interleaved layout (KNHIP_AOS_KEEP_MB overrides; tests
// set it to 0 to exercise the rebuild path)
size_t aos_keep_limit() { return env_layout().aos_keep_bytes; }

// lay the lists out from device-resident, list-sorted AoS codes + ids
int build_list_layout(knhip_index* idx, const std::vector<int64_t>& list_off, const uint8_t* d_codes,
                      const int64_t* d_ids) {
    const int64_t nlist = idx->nlist;
    const int64_t ntotal = list_off[nlist];
    idx->ntotal = ntotal;
    idx->row_scale.release(); // (stored norms belong to one layout: the caller sets them again after an Add)
    idx->cos_mode = 0;
    idx->h_list_len.resize(nlist);
    idx->h_list_row_off.assign(list_off.begin(), list_off.begin() + nlist);
    idx->max_list_len = 0;
    {
        const EnvLayout env = env_layout(); // (the switches an index keeps for its lifetime: knhip_env.h)
        idx->rank0_select = env.rank0_select;
        idx->cand_hist = env.cand_hist;
        idx->pq_q4 = env.pq_q4;
        idx->mscan = env.mscan;
        idx->flat_bf16 = env.flat_bf16;
        idx->mscan_cap = env.mscan_cap;
        idx->pqd_spill_cap = env.pqd_spill_cap;
        idx->xnorm_ready = false;
        idx->pqf = env.pqf;
        idx->pqf_guard = env.pqf_guard;
        idx->pqf_form = env.pqf_form;
        idx->pq_v1 = env.pq_v1;
        idx->psum_ready = false;
        idx->pqf_ready = false;
        idx->pqi_ready = false;
        idx->pqd_ready = false;
        idx->idmap_ready = false;
        idx->rg_seg_nseg = -1;
        idx->idmap_ids.release();
        idx->idmap_col.release();
        idx->rows_i.release();
        idx->rows_r.release();
        idx->psum.release();
        idx->psum_s.release();
    }
    for (int64_t l = 0; l < nlist; l++) {
        idx->h_list_len[l] = list_off[l + 1] - list_off[l];
        idx->max_list_len = std::max(idx->max_list_len, idx->h_list_len[l]);
        if (idx->h_list_len[l] < 0) {
            return fail(KNHIP_ERR_INVALID_ARGS, "list offsets must be non-decreasing");
        }
    }
    std::vector<int64_t> blk_off(nlist + 1, 0);
    const int kind = idx->desc.kind;
    for (int64_t l = 0; l < nlist; l++) {
        int64_t nb;
        if (kind == KNHIP_IVF_PQ) {
            nb = pq_skew_blocks(idx->h_list_len[l], idx->desc.pq_m);
        } else {
            nb = (idx->h_list_len[l] + 63) / 64;
        }
        blk_off[l + 1] = blk_off[l] + nb;
    }
    int rc;
    if ((rc = upload(idx->d_list_len, idx->h_list_len.data(), nlist * sizeof(int64_t)))) return rc;
    if ((rc = upload(idx->d_list_row_off, idx->h_list_row_off.data(), nlist * sizeof(int64_t)))) return rc;
    if ((rc = upload(idx->d_list_blk_off, blk_off.data(), (nlist + 1) * sizeof(int64_t)))) return rc;
    idx->h_list_off = list_off;
    if (idx->ids.p != d_ids) {
        HIP_TRY(idx->ids.alloc((size_t)ntotal * sizeof(int64_t)));
        if (ntotal) {
            HIP_TRY(hipMemcpy(idx->ids.p, d_ids, (size_t)ntotal * sizeof(int64_t), hipMemcpyDeviceToDevice));
        }
    }
    // the canonical AoS bytes stay resident for IVF_PQ (they feed the lazily built layouts; 32 B / row) and for small
    // flat / SQ8 indexes; large flat / SQ8 lists keep only the interleaved layout (C5: 76.8 GB of codes once, not twice)
    const size_t aos_bytes = (size_t)ntotal * idx->code_size;
    const bool keep_aos = kind == KNHIP_IVF_PQ || aos_bytes <= aos_keep_limit();
    if (idx->codes_aos.p != d_codes) {
        idx->codes_aos.release();
        if (keep_aos) {
            HIP_TRY(idx->codes_aos.alloc(aos_bytes));
            if (ntotal) {
                HIP_TRY(hipMemcpy(idx->codes_aos.p, d_codes, aos_bytes, hipMemcpyDeviceToDevice));
            }
        }
    }
    idx->aos_ready = keep_aos;
    const int64_t total_blk = blk_off[nlist];
    idx->total_blk = total_blk;
    idx->h_list_blk_off = blk_off;
    if (kind == KNHIP_IVF_FLAT) {
        const int nchunk = (idx->d + 3) / 4;
        HIP_TRY(idx->rows.alloc((size_t)total_blk * nchunk * 64 * sizeof(float4)));
        HIP_TRY(launch_interleave_lists(reinterpret_cast<const float*>(d_codes),
                                        idx->d_list_row_off.as<int64_t>(), idx->d_list_len.as<int64_t>(),
                                        idx->d_list_blk_off.as<int64_t>(), nlist, idx->d,
                                        idx->rows.as<float4>(), nullptr));
    } else if (kind == KNHIP_IVF_PQ) {
        const int M = idx->desc.pq_m;
        idx->pq_v2 = (M == 32) && !idx->pq_v1;
        idx->rows.release();
        idx->skew_ready = false;
        if (!idx->pq_v2 && pq_scan_supported_m(M)) {
            // (m = 32 searches on the stream16 layout; its skewed copy is built on first use: k > 128.  Widths without a
            // systolic kernel -- pq_scan_any.hip -- read the canonical AoS codes)
            if (int rc2 = build_pq_skew(idx)) return rc2;
        }
        if (idx->pq_v2) {
            std::vector<int64_t> off2(nlist + 1, 0);
            for (int64_t l = 0; l < nlist; l++) {
                off2[l + 1] = off2[l] + pq_stream16_blocks(idx->h_list_len[l]);
            }
            if ((rc = upload(idx->d_list_blk_off2, off2.data(), (nlist + 1) * sizeof(int64_t)))) return rc;
            HIP_TRY(idx->rows2.alloc((size_t)off2[nlist] * 64 * sizeof(uint4)));
            HIP_TRY(launch_pq_stream16(d_codes, idx->d_list_row_off.as<int64_t>(), idx->d_list_len.as<int64_t>(),
                                       idx->d_list_blk_off2.as<int64_t>(), nlist, idx->rows2.as<uint4>(), nullptr));
        }
    } else if (kind == KNHIP_IVF_SQ8) {
        const int nchunk16 = (idx->d + 15) / 16;
        HIP_TRY(idx->rows.alloc((size_t)total_blk * nchunk16 * 64 * sizeof(uint4)));
        HIP_TRY(launch_sq_interleave(d_codes, idx->d_list_row_off.as<int64_t>(),
                                     idx->d_list_len.as<int64_t>(), idx->d_list_blk_off.as<int64_t>(),
                                     nlist, idx->d, idx->rows.as<uint4>(), nullptr));
    } else {
        return fail(KNHIP_ERR_INVALID_ARGS, "lists on a brute-force index");
    }
    HIP_TRY(hipDeviceSynchronize());
    if (!keep_aos) {
        idx->codes_aos.release(); // (d_codes may have been this buffer: the layout above was built from it first)
    }
    idx->has_data = true;
    return KNHIP_OK;
}

// canonical list-sorted AoS codes of a flat / SQ8 index that dropped them: rebuilt from the interleaved blocks
int ensure_aos(const knhip_index* cidx) {
    knhip_index* idx = const_cast<knhip_index*>(cidx);
    if (idx->aos_ready || !idx->has_data || idx->desc.kind == KNHIP_BRUTE_FORCE) {
        return KNHIP_OK;
    }
    HIP_TRY(idx->codes_aos.alloc((size_t)std::max<int64_t>(idx->ntotal, 1) * idx->code_size));
    HIP_TRY(launch_deinterleave_lists(idx->rows.as<uint4>(), idx->d_list_row_off.as<int64_t>(),
                                      idx->d_list_len.as<int64_t>(), idx->d_list_blk_off.as<int64_t>(), idx->nlist,
                                      idx->code_size, idx->codes_aos.as<uint8_t>(), nullptr));
    HIP_TRY(hipDeviceSynchronize());
    idx->aos_ready = true;
    return KNHIP_OK;
}

// ||x||^2 per stored row (fp32 rows: any metric, for the error bound; SQ8: L2 only), built on first use
int ensure_mscan_norms(const knhip_index* idx) {
    const int64_t total_blk = idx->total_blk;
    std::lock_guard<std::mutex> lk(idx->mu);
    if (idx->xnorm_ready) {
        return KNHIP_OK;
    }
    HIP_TRY(idx->xnorm.alloc((size_t)std::max<int64_t>(total_blk, 1) * 64 * sizeof(float) + 16));
    float* xn = idx->xnorm.as<float>();
    float* xmax = xn + std::max<int64_t>(total_blk, 1) * 64;
    if (idx->desc.kind == KNHIP_IVF_FLAT) {
        HIP_TRY(launch_ms_block_norms(idx->rows.as<float4>(), total_blk, (idx->d + 3) / 4, xn, xmax, nullptr));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(&idx->xnorm_max, xmax, sizeof(float), hipMemcpyDeviceToHost));
    } else {
        HIP_TRY(launch_ms_sq8_norms(idx->rows.as<uint4>(), total_blk, (idx->d + 15) / 16, idx->d,
                                    idx->sq_trained.as<float>(), xn, xmax, nullptr));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(&idx->xnorm_max, xmax, sizeof(float), hipMemcpyDeviceToHost));
    }
    idx->xnorm_ready = true;
    return KNHIP_OK;
}

constexpr int KNHIP_PQF_ABANDONED = 1; // (not an error: the selectivity guard sent the batch to the exact kernel)

// IVF-PQ prefilter, every form and the sample pass: per-vector term-2 sums (+ the block offsets they are indexed by), from the
// canonical AoS codes
int ensure_psum(const knhip_index* idx) {
    std::lock_guard<std::mutex> lk(idx->mu);
    if (idx->psum_ready) {
        return KNHIP_OK;
    }
    const int64_t nlist = idx->nlist;
    std::vector<int64_t> off(nlist + 1, 0);
    for (int64_t l = 0; l < nlist; l++) {
        off[l + 1] = off[l] + pq_stream16r_blocks(idx->h_list_len[l]);
    }
    HIP_TRY(idx->d_list_blk_off_r.alloc((size_t)(nlist + 1) * sizeof(int64_t)));
    HIP_TRY(hipMemcpy(idx->d_list_blk_off_r.p, off.data(), (size_t)(nlist + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    idx->pabs_max = 0.f;
    if (idx->is_l2) {
        const size_t npos = (size_t)std::max<int64_t>(off[nlist], 1) * 16; // 16 vector positions per block
        HIP_TRY(idx->psum.alloc((npos + 4) * sizeof(float)));
        HIP_TRY(hipMemset(idx->psum.p, 0, (npos + 4) * sizeof(float)));
        uint32_t* bits = reinterpret_cast<uint32_t*>(idx->psum.as<float>() + npos);
        HIP_TRY(launch_pq_psum(idx->codes_aos.as<uint8_t>(), idx->d_list_row_off.as<int64_t>(),
                               idx->d_list_len.as<int64_t>(), idx->d_list_blk_off_r.as<int64_t>(), nlist,
                               idx->precomp_t.as<float>(), idx->psum.as<float>(), bits, nullptr));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(&idx->pabs_max, bits, sizeof(float), hipMemcpyDeviceToHost));
    }
    HIP_TRY(hipDeviceSynchronize());
    idx->psum_ready = true;
    return KNHIP_OK;
}

// ... the half form: + the rotated token stream
int ensure_pqf(const knhip_index* idx) {
    if (int rc = ensure_psum(idx)) return rc;
    std::lock_guard<std::mutex> lk(idx->mu);
    if (idx->pqf_ready) {
        return KNHIP_OK;
    }
    const int64_t nblk = idx->nlist > 0 ? [&] {
        int64_t t = 0;
        for (int64_t l = 0; l < idx->nlist; l++) t += pq_stream16r_blocks(idx->h_list_len[l]);
        return t;
    }() : 0;
    HIP_TRY(idx->rows_r.alloc((size_t)std::max<int64_t>(nblk, 1) * 64 * sizeof(uint4)));
    HIP_TRY(launch_pq_stream16r(idx->codes_aos.as<uint8_t>(), idx->d_list_row_off.as<int64_t>(),
                                idx->d_list_len.as<int64_t>(), idx->d_list_blk_off_r.as<int64_t>(), idx->nlist,
                                idx->rows_r.as<uint4>(), nullptr));
    HIP_TRY(hipDeviceSynchronize());
    idx->pqf_ready = true;
    return KNHIP_OK;
}

// ... the decode form (pq_decode.hip): the codebook as halves, the index's scales, the rows' start values -psum SC / 2.  No
// token stream: the kernel reads the canonical AoS codes as they lie
int ensure_pqd(const knhip_index* idx) {
    if (int rc = ensure_psum(idx)) return rc;
    std::lock_guard<std::mutex> lk(idx->mu);
    if (idx->pqd_ready) {
        return KNHIP_OK;
    }
    HIP_TRY(idx->pqd_cb16.alloc((size_t)32 * 256 * 4 * 2));
    HIP_TRY(idx->pqd_st.alloc(16 * sizeof(float)));
    const int64_t npsum = idx->is_l2 ? (int64_t)(idx->psum.bytes / sizeof(float)) - 4 : 0;
    if (npsum > 0) {
        // (+ 64: the scan's last 32-row tile of the last list reads up to 16 entries past the list's 16-row blocks)
        HIP_TRY(idx->psum_s.alloc((size_t)(npsum + 64) * sizeof(float)));
        HIP_TRY(hipMemset(idx->psum_s.p, 0, (size_t)(npsum + 64) * sizeof(float)));
    }
    HIP_TRY(launch_pqd_index_prep(idx->cb.as<float4>(), idx->centroids.as<float>(), idx->nlist * (int64_t)idx->d,
                                  idx->pqd_cb16.p, idx->pqd_st.as<float>(),
                                  reinterpret_cast<uint32_t*>(idx->pqd_st.as<float>() + 8), idx->psum.as<float>(), npsum,
                                  idx->psum_s.as<float>(), nullptr));
    HIP_TRY(hipDeviceSynchronize());
    idx->pqd_ready = true;
    return KNHIP_OK;
}

// token stream of the integer form of the prefilter (same block offsets as the half-precision stream)
int ensure_pqi(const knhip_index* idx) {
    if (int rc = ensure_pqf(idx)) return rc;
    std::lock_guard<std::mutex> lk(idx->mu);
    if (idx->pqi_ready) {
        return KNHIP_OK;
    }
    const int64_t nblk = idx->rows_r.bytes / (64 * sizeof(uint4));
    HIP_TRY(idx->rows_i.alloc((size_t)std::max<int64_t>(nblk, 1) * 64 * sizeof(uint4)));
    HIP_TRY(launch_pq_stream16i(idx->codes_aos.as<uint8_t>(), idx->d_list_row_off.as<int64_t>(),
                                idx->d_list_len.as<int64_t>(), idx->d_list_blk_off_r.as<int64_t>(), idx->nlist,
                                idx->rows_i.as<uint4>(), nullptr));
    HIP_TRY(hipDeviceSynchronize());
    idx->pqi_ready = true;
    return KNHIP_OK;
}

Workspace* acquire_ws(const knhip_index* idx, void* stream_key, bool pooled_by_stream) {
    std::lock_guard<std::mutex> lk(idx->mu);
    if (pooled_by_stream) {
        auto& slot = idx->ws_by_stream[stream_key];
        if (!slot) {
            slot.reset(new Workspace());
        }
        return slot.get();
    }
    if (!idx->ws_free.empty()) {
        Workspace* w = idx->ws_free.back().release();
        idx->ws_free.pop_back();
        return w;
    }
    return new Workspace();
}

void release_ws(const knhip_index* idx, Workspace* w) {
    std::lock_guard<std::mutex> lk(idx->mu);
    idx->ws_free.emplace_back(w);
}

// ---- BRUTE_FORCE on the matrix cores ---------------------------------------------------------------------------------------
// The reference's batch path of BruteForce / IndexFlat::search is a dense contraction too (exhaustive_L2sqr_blas,
// thirdparty/faiss/faiss/cppcontrib/knowhere/utils/distances.cpp:994-1050: sgemm + norms, reservoir / heap per block); its
// one-query-per-task path -- what Knowhere's nodes drive, src/common/comp/brute_force.cc:258-392 -- is the sequential
// fvec_L2sqr / fvec_inner_product per row, whose bits the exact row scan (flat_scan.hip) reproduces.  Here the coarse
// quantizer's machinery runs over the BASE rows: split-bf16 prefilter on the matrix pipe with the selection fused
// (coarse_gemm.hip: two GEMM passes, no nq x nb matrix), the candidates recomputed in the reference's sequential order,
// a certificate that nothing unselected can beat the k-th (else the query is redone by the exact all-pairs kernel) -- the
// same bits and the same canonical order as the row scan, at matrix-pipe speed.  The base is cut into chunks of at most
// 131072 rows (the bound kernel holds up to 4096 group minima per query); the chunks' top-k lists are merged as the row
// scan's are.  Taken when no bitset is given (the prefilter cannot count filtered rows), the metric is plain L2 / IP, and the
// batch is large enough to pay for splitting the queries.
int ensure_bf_split(const knhip_index* idx) {
    std::lock_guard<std::mutex> lk(idx->mu);
    if (idx->bf_split_ready) {
        return KNHIP_OK;
    }
    const int64_t nb = idx->ntotal;
    const int nslab = coarse_bf16_slabs(idx->d);
    HIP_TRY(idx->rows_bs.alloc((size_t)std::max<int64_t>(nb, 1) * nslab * 128));
    HIP_TRY(idx->bf_norm.alloc((size_t)std::max<int64_t>(nb, 1) * sizeof(float)));
    HIP_TRY(launch_coarse_bf16_split(idx->codes_aos.as<float>(), nb, idx->d, idx->rows_bs.p, nullptr));
    HIP_TRY(launch_row_norms(idx->codes_aos.as<float>(), nb, idx->d, idx->bf_norm.as<float>(), nullptr));
    HIP_TRY(hipDeviceSynchronize());
    std::vector<float> h((size_t)nb);
    HIP_TRY(hipMemcpy(h.data(), idx->bf_norm.p, h.size() * sizeof(float), hipMemcpyDeviceToHost));
    float mx = 0.f;
    for (float v : h) {
        mx = v > mx ? v : mx; // (a NaN norm never raises it: such a row fails every certificate it could decide)
    }
    idx->bf_norm_max = mx;
    idx->bf_split_ready = true;
    return KNHIP_OK;
}

// a chunk's (nq, k) result -> slot `slot` of the partial lists [nq][nslots][k], row numbers -> ids
__global__ void bf_place_kernel(const int64_t* __restrict__ keys, const float* __restrict__ dis, int64_t nq, int k, int64_t add,
                                float* __restrict__ pd, int64_t* __restrict__ pi, int nslots, int slot) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nq * k) {
        return;
    }
    const int64_t q = t / k, j = t % k;
    const int64_t o = (q * nslots + slot) * k + j;
    const int64_t key = keys[t];
    pd[o] = dis[t];
    pi[o] = key >= 0 ? key + add : -1;
}

// rows of a chunk (multiples of 128), 0 = the shape is not served
static int64_t bf_mfma_chunk_rows(int64_t nb, int k) {
    const int ncand = k + std::max(32, k / 4);
    const int64_t nch = (nb + 131071) / 131072;
    const int64_t per = round_up((nb + nch - 1) / nch, 128);
    const int64_t last = nb - (nch - 1) * per;
    if (last <= 0 || !coarse_bf16_supports(per, ncand) || !coarse_bf16_supports(last, ncand) || ncand >= last) {
        return 0;
    }
    return per;
}

int bf_mfma_batch(const knhip_index* idx, Workspace* ws, const float* d_q, int64_t nq, int k, int64_t per, int64_t* d_out_i,
                  float* d_out_d, hipStream_t s) {
    if (int rc = ensure_bf_split(idx)) return rc;
    const int64_t nb = idx->ntotal;
    const int d = idx->d;
    const int nslab = coarse_bf16_slabs(d);
    const int64_t nch = (nb + per - 1) / per;
    const int nchunk4 = (d + 3) / 4;
    // queries per round: the exact fallback's nq x rows scratch stays below 1 GiB
    const int64_t nqb = std::max<int64_t>(64, std::min<int64_t>(nq, ((int64_t)1 << 28) / per));
    HIP_TRY(ws->partial_d.reserve((size_t)nq * nch * k * sizeof(float)));
    HIP_TRY(ws->partial_i.reserve((size_t)nq * nch * k * sizeof(int64_t)));
    HIP_TRY(ws->keys.reserve((size_t)nqb * k * sizeof(int64_t)));
    HIP_TRY(ws->cdis.reserve((size_t)nqb * k * sizeof(float)));
    {
        StageTimer t(idx, s, KNHIP_STAGE_SCAN);
        HIP_TRY(ws->bf_kth.reserve((size_t)nqb * sizeof(float)));
        for (int64_t q0 = 0; q0 < nq; q0 += nqb) {
            const int64_t n = std::min(nqb, nq - q0);
            // (one pass over every chunk but the first: the k-th best of the chunks searched so far is the selection bound)
            HIP_TRY(launch_fill_f32(ws->bf_kth.as<float>(), n, idx->is_l2 ? FLT_MAX : -FLT_MAX, s));
            RowsRun run{ws->bf_kth.as<float>(), false, false};
            for (int64_t c = 0; c < nch; c++) {
                const int64_t r0 = c * per, rn = std::min(per, nb - r0);
                CoarseRows R{idx->codes_aos.as<float>() + r0 * d, idx->rows.as<float4>() + (r0 / 64) * nchunk4 * 64,
                             static_cast<const unsigned char*>(idx->rows_bs.p) + (size_t)r0 * nslab * 128,
                             idx->bf_norm.as<float>() + r0, idx->bf_norm_max, rn};
                if (int rc = coarse_rows_stage(idx, ws, R, d_q + q0 * d, n, k, ws->keys.as<int64_t>(), ws->cdis.as<float>(), s,
                                               &run)) {
                    return rc;
                }
                hipLaunchKernelGGL(bf_place_kernel, dim3((unsigned)((n * k + 255) / 256)), dim3(256), 0, s, ws->keys.as<int64_t>(),
                                   ws->cdis.as<float>(), n, k, r0 + idx->id_offset, ws->partial_d.as<float>() + q0 * nch * k,
                                   ws->partial_i.as<int64_t>() + q0 * nch * k, (int)nch, (int)c);
                HIP_TRY(hipGetLastError());
            }
        }
    }
    {
        std::lock_guard<std::mutex> lk(idx->mu);
        idx->coarse_flops += 0.0; // (the stage's flop count is the bench's: 2 nq nb d)
    }
    StageTimer t(idx, s, KNHIP_STAGE_MERGE);
    HIP_TRY(launch_merge_partials(ws->partial_d.as<float>(), ws->partial_i.as<int64_t>(), nq, (int)nch, k, nch * k, k,
                                  idx->is_l2, d_out_d, d_out_i, s));
    return KNHIP_OK;
}

// ---- one batch of queries, everything on the device ------------------------------------------------
// pre_keys / pre_cdis non-null: the coarse assignment is given (IndexIVF::search_preassigned), [nq][nprobe]
int search_batch(const knhip_index* idx, Workspace* ws, const float* d_q, int64_t nq, int k, int nprobe,
                 const uint8_t* d_bitset, int64_t nbits, int64_t* d_out_i, float* d_out_d,
                 hipStream_t s, const int64_t* pre_keys, const float* pre_cdis) {
    const int kind = idx->desc.kind;
    const int d = idx->d;
    const bool is_l2 = idx->is_l2;
    HIP_TRY(ws->gthr.reserve((size_t)nq * sizeof(float)));
    HIP_TRY(launch_fill_f32(ws->gthr.as<float>(), nq, is_l2 ? FLT_MAX : -FLT_MAX, s));

    if (kind == KNHIP_BRUTE_FORCE) {
        const int64_t nb = idx->ntotal;
        if (idx->bf_mfma && d_bitset == nullptr && idx->cos_mode == 0 && idx->coarse_gemm == 2 && nb >= 4096 && nq >= 16 &&
            (double)nq * (double)nb >= 16.0e6) {
            const int64_t per = bf_mfma_chunk_rows(nb, k);
            if (per > 0) {
                idx->last_bf_mfma = 1;
                return bf_mfma_batch(idx, ws, d_q, nq, k, per, d_out_i, d_out_d, s);
            }
        }
        idx->last_bf_mfma = 0;
        int64_t chunk_rows = std::max<int64_t>(1024, round_up((nb + 511) / 512, 64));
        const int64_t nchunks = (nb + chunk_rows - 1) / chunk_rows;
        const int qg = flat_scan_qg(k);
        const int64_t ngroups = (nq + qg - 1) / qg;
        HIP_TRY(ws->partial_d.reserve((size_t)nq * nchunks * k * sizeof(float)));
        HIP_TRY(ws->partial_i.reserve((size_t)nq * nchunks * k * sizeof(int64_t)));
        FlatScanArgs a{};
        a.rows = idx->rows.as<float4>();
        a.nrows = nb;
        a.chunk_rows = chunk_rows;
        a.id_offset = idx->id_offset;
        a.d = d;
        a.nchunk = (d + 3) / 4;
        a.queries = d_q;
        a.nq = nq;
        a.nitems_dense = nchunks * ngroups;
        a.ngroups = ngroups;
        a.bitset = d_bitset;
        a.bitset_nbits = nbits;
        a.partial_d = ws->partial_d.as<float>();
        a.partial_i = ws->partial_i.as<int64_t>();
        a.gthr = ws->gthr.as<float>();
        a.nslot = (int)nchunks;
        a.k = k;
        a.row_scale = idx->row_scale.as<float>();
        a.cos_mode = idx->cos_mode;
        {
            StageTimer t(idx, s, KNHIP_STAGE_SCAN);
            HIP_TRY(launch_flat_scan(a, is_l2, true, a.nitems_dense, s));
        }
        {
            StageTimer t(idx, s, KNHIP_STAGE_MERGE);
            HIP_TRY(launch_merge_partials(a.partial_d, a.partial_i, nq, (int)nchunks, k, nchunks * k, k,
                                          is_l2, d_out_d, d_out_i, s));
        }
        return KNHIP_OK;
    }

    // ---- IVF kinds ----
    const int64_t nlist = idx->nlist;
    // 1. coarse
    const int64_t* keys_p = pre_keys;
    const float* cdis_p = pre_cdis;
    if (pre_keys == nullptr) {
        HIP_TRY(ws->keys.reserve((size_t)nq * nprobe * sizeof(int64_t)));
        HIP_TRY(ws->cdis.reserve((size_t)nq * nprobe * sizeof(float)));
        StageTimer t(idx, s, KNHIP_STAGE_COARSE);
        if (int rc = coarse_stage(idx, ws, d_q, nq, nprobe, ws->keys.as<int64_t>(), ws->cdis.as<float>(), s)) {
            return rc;
        }
        keys_p = ws->keys.as<int64_t>();
        cdis_p = ws->cdis.as<float>();
    }
    if (kind == KNHIP_IVF_PQ && !pq_scan_supported_m(idx->desc.pq_m)) {
        // any other number of sub-quantizers: the plain exact kernel, one workgroup per (query, probe), no work table
        const int M = idx->desc.pq_m;
        const int mode = !is_l2 ? PQ_LUT_IP : (idx->use_precomp ? PQ_LUT_PRECOMP : PQ_LUT_RESIDUAL);
        if (mode != PQ_LUT_RESIDUAL) {
            HIP_TRY(ws->t2t.reserve((size_t)nq * 256 * M * sizeof(float)));
            StageTimer t(idx, s, KNHIP_STAGE_LUT);
            HIP_TRY(launch_pq_query_table(d_q, idx->cb.as<float>(), d, M, nq, ws->t2t.as<float>(), s));
        }
        const int64_t nparts = (int64_t)nprobe * pq_scan_any_parts(k);
        HIP_TRY(ws->partial_d.reserve((size_t)nq * nparts * k * sizeof(float)));
        HIP_TRY(ws->partial_i.reserve((size_t)nq * nparts * k * sizeof(int64_t)));
        PqAnyArgs a{};
        a.keys = keys_p;
        a.coarse_dis = cdis_p;
        a.nprobe = nprobe;
        a.nlist = nlist;
        a.list_len = idx->d_list_len.as<int64_t>();
        a.list_row_off = idx->d_list_row_off.as<int64_t>();
        a.codes = idx->codes_aos.as<uint8_t>();
        a.ids = idx->ids.as<int64_t>();
        a.M = M;
        a.d = d;
        a.lut_mode = mode;
        a.t2t = ws->t2t.as<float>();
        a.precomp_t = idx->precomp_t.as<float>();
        a.cb = idx->cb.as<float>();
        a.centroids = idx->centroids.as<float>();
        a.queries = d_q;
        a.bitset = d_bitset;
        a.bitset_nbits = nbits;
        a.partial_d = ws->partial_d.as<float>();
        a.partial_i = ws->partial_i.as<int64_t>();
        a.k = k;
        {
            StageTimer t(idx, s, KNHIP_STAGE_SCAN);
            HIP_TRY(launch_pq_scan_any(a, nq, is_l2, s));
        }
        StageTimer t(idx, s, KNHIP_STAGE_MERGE);
        HIP_TRY(launch_merge_partials(a.partial_d, a.partial_i, nq, (int)nparts, k, nparts * k, k, is_l2, d_out_d, d_out_i, s));
        return KNHIP_OK;
    }
    // 2. group
    const int qg = (kind == KNHIP_IVF_PQ) ? pq_scan_qg(idx->desc.pq_m)
                 : (kind == KNHIP_IVF_SQ8) ? sq_scan_qg(k)
                                           : flat_scan_qg(k);
    const int64_t npairs = nq * nprobe;
    // IVF-PQ m = 32: which kernels run the two phases (rank-0 dump + select, bulk) and how many queries they
    // take per work item
    bool pq_use_v2 = false, pq_rank0 = false, pq_use_q4 = false;
    int qg_rank0 = qg, qg_bulk = qg;
    if (kind == KNHIP_IVF_PQ && idx->pq_v2 && pq_scan_v2_supports(idx->desc.pq_m, k)) {
        pq_use_v2 = true;
        const int64_t stride = round_up(std::max<int64_t>(idx->max_list_len, 64), 64);
        // (worth it once k is large enough that sorted insertion dominates: measured k >= 32)
        pq_rank0 = idx->rank0_select && k >= 32 && nprobe > 1 && (double)nq * stride * 4.0 <= 6.0e9;
        // the 4-query kernel pays when the lists are shared by enough (query, probe) pairs of the batch
        pq_use_q4 = idx->cb_t.p != nullptr && pq_scan_q4_supports(idx->desc.pq_m, d, k) &&
                (idx->pq_q4 == 1 || (idx->pq_q4 == 2 && npairs >= 6 * nlist));
        if (pq_use_q4) {
            qg_bulk = 4;
            qg_rank0 = pq_rank0 ? qg : 4;
        }
    }
    // IVF-Flat / IVF-SQ8: MFMA prefilter + exact finish (mfma_scan.hip) when the lists are shared by enough queries.
    // IVF-PQ m = 32: the matrix-core ADC prefilter (pq_filter.hip) through the same machinery, when the lists are shared by
    // enough queries for its units of (list, 8 queries) -- 4 pairs per list on average --; its exact fallback is the
    // 4-query kernel.
    bool use_ms = false;
    int ms_cap = 0, ms_nchunk = 0, ms_nstep = 0;
    // (k <= 128: the exact fallback of its overflowed queries is the 4-query kernel; 128 < k <= 1024 -- Knowhere's refine
    // asks for k * refine_k candidates -- it is the systolic kernel over one-pair items: the filter, the sample and the
    // finish take any k)
    const bool pq_q4_ok = kind == KNHIP_IVF_PQ && pq_scan_q4_supports(idx->desc.pq_m, d, k);
    const bool pqf_shape = kind == KNHIP_IVF_PQ && idx->pqf != 0 && idx->pq_v2 && idx->cb_t.p != nullptr &&
            pqf_supports(idx->desc.pq_m, d) && k <= 1024 && (pq_q4_ok || pq_scan_supported_m(idx->desc.pq_m)) &&
            (!is_l2 || idx->use_precomp); // (residual tables: see pq_psum_kernel)
    // (COSINE with stored norms takes the exact kernels: the prefilter's bound does not carry the per-row division)
    if ((kind == KNHIP_IVF_FLAT || kind == KNHIP_IVF_SQ8 || pqf_shape) && (idx->mscan != 0 || pqf_shape) && nprobe >= 2 &&
        idx->cos_mode == 0) {
        size_t lds;
        if (kind == KNHIP_IVF_PQ) {
            ms_nchunk = d / 4;
            lds = pqf_smem();
        } else if (kind == KNHIP_IVF_FLAT) {
            ms_nchunk = (d + 3) / 4;
            ms_nstep = (ms_nchunk + 3) / 4;
            lds = mscan_flat_smem(ms_nstep);
        } else {
            ms_nchunk = (d + 15) / 16;
            ms_nstep = (ms_nchunk + 1) / 2;
            lds = mscan_sq8_smem(ms_nstep);
        }
        // candidate capacity per query: the finish kernel sorts them in LDS (a power of two entries)
        // candidate capacity per query (the finish kernel takes any number, in chunks): generous -- a query whose sample
        // gave a loose bound collects thousands of rows before its histogram tightens it, and their exact distances
        // cost far less than the exact scan of all its lists -- within ~3 GB of scratch per batch
        ms_cap = 4096;
        while (ms_cap < (int64_t)16 * nprobe * k && ms_cap < 32768) {
            ms_cap <<= 1;
        }
        while (ms_cap > 1024 && (double)ms_cap * (double)nq * 8.0 > 3.0e9) {
            ms_cap >>= 1;
        }
        if (idx->mscan_cap > 0) {
            ms_cap = idx->mscan_cap;
        }
        use_ms = lds <= 160 * 1024 - 1024 && ms_cap >= 2 * k &&
                (kind == KNHIP_IVF_PQ ? (idx->pqf == 2 || npairs >= 4 * nlist) : (idx->mscan == 1 || npairs >= 8 * nlist));
    }
    if (use_ms && kind == KNHIP_IVF_PQ) { // (no rank-0 dump phase: the sample pass of the prefilter gives the bounds)
        pq_rank0 = false;
        pq_use_q4 = pq_q4_ok;
        qg_bulk = pq_q4_ok ? 4 : qg;
        qg_rank0 = pq_q4_ok ? 4 : qg;
    }
    const int64_t items_bound =
            round_up(npairs / std::min(qg_rank0, qg_bulk) + std::min<int64_t>(2 * nlist, npairs) + 1, 8);
    HIP_TRY(ws->list_count.reserve((size_t)2 * nlist * sizeof(int32_t)));
    HIP_TRY(ws->list_cursor.reserve((size_t)2 * nlist * sizeof(int32_t)));
    HIP_TRY(ws->list_pair_off.reserve((size_t)(2 * nlist + 1) * sizeof(int64_t)));
    HIP_TRY(ws->list_item_off.reserve((size_t)(2 * nlist + 1) * sizeof(int64_t)));
    HIP_TRY(ws->pairs.reserve((size_t)npairs * sizeof(KnPair)));
    // (the MFMA prefilter's fallback compacts the pairs of overflowed queries into one-query items: up to npairs)
    HIP_TRY(ws->items.reserve((size_t)(use_ms ? std::max<int64_t>(npairs, items_bound) : items_bound) * sizeof(KnItem)));
    HIP_TRY(ws->nitems.reserve(sizeof(int64_t)));
    // rows of a query's sample (IVF-Flat / IVF-SQ8 prefilter), at most: the first max(1024, 8 k) rows of its closest
    // list(s) -- the pass is bound by the rows it reads (C2: every list is somebody's closest: the whole index once per
    // batch when a list was sampled in full), and tau from 1024 rows lets only a few dozen more candidates through.
    // KNHIP_MS_SAMPLE_ROWS=n overrides (tests / experiments; 8192 = whole lists as in rounds 2-4)
    // (IVF-SQ8 too, now that its finish prunes: before that the looser tau cost C5's finish 2.7 ms for 1.1 ms saved here)
    int ms_sample_cap = std::min<int>(mscan_sample_rows(), (std::max(1024, 8 * k) + 63) / 64 * 64);
    const EnvSearch env = env_search(); // (the switches every search reads: knhip_env.h)
    if (env.ms_sample_rows > 0) {
        ms_sample_cap = std::max(64, std::min(mscan_sample_rows(), env.ms_sample_rows / 64 * 64));
    }
    WorkTable wt{};
    wt.list_count = ws->list_count.as<int32_t>();
    wt.list_cursor = ws->list_cursor.as<int32_t>();
    wt.list_pair_off = ws->list_pair_off.as<int64_t>();
    wt.list_item_off = ws->list_item_off.as<int64_t>();
    wt.pairs = ws->pairs.as<KnPair>();
    wt.items = ws->items.as<KnItem>();
    wt.nitems = ws->nitems.as<int64_t>();
    wt.scan_bytes = idx->scan_bytes_dev.as<double>();
    HIP_TRY(ws->partial_d.reserve((size_t)npairs * k * sizeof(float)));
    HIP_TRY(ws->partial_i.reserve((size_t)npairs * k * sizeof(int64_t)));
    wt.empty_mark = ws->partial_i.as<int64_t>();
    wt.k = k;
    // The IVF-PQ prefilter samples per query (pq_filter.hip, pq_sample_kernel) and groups all probes of a list together
    // afterwards: it needs this table -- split by the sample plan -- only when its guard abandons the batch.
    const bool wt1_lazy = use_ms && kind == KNHIP_IVF_PQ;
    auto build_wt1 = [&]() -> int {
        StageTimer t(idx, s, KNHIP_STAGE_GROUP);
        const int32_t* cls = nullptr;
        if (use_ms) {
            // the first class of the split = the pairs whose rows feed tau_q (mfma_scan.hip sample plan): the probes in
            // coarse order until max(1024, 8 k) rows are covered
            HIP_TRY(ws->ms_sample_off.reserve((size_t)npairs * sizeof(int32_t)));
            HIP_TRY(ws->ms_nrow.reserve((size_t)nq * sizeof(int32_t)));
            HIP_TRY(launch_ms_sample_plan(keys_p, nq, nprobe, nlist, idx->d_list_len.as<int64_t>(),
                                          std::max(1024, 8 * k), ms_sample_cap, ws->ms_sample_off.as<int32_t>(),
                                          ws->ms_nrow.as<int32_t>(), s));
            cls = ws->ms_sample_off.as<int32_t>();
        }
        HIP_TRY(launch_build_worktable(keys_p, nq, nprobe, nlist, qg_rank0, qg_bulk,
                                       idx->d_list_len.as<int64_t>(), idx->code_size, wt, s, 0, cls));
        return KNHIP_OK;
    };
    if (!wt1_lazy) {
        if (int rc = build_wt1()) return rc;
    }
    {
        std::lock_guard<std::mutex> lk(idx->mu);
        idx->last_items_bound = items_bound;
    }

    // The MFMA prefilter path (mfma_scan.hip): sample -> tau_q, filter, exact finish; `exact_one` runs the exact kernel
    // over a compact table of one-query items (the overflowed queries; normally none, the kernels then return at once)
    auto run_mscan = [&](const std::function<int(const KnItem*, const KnPair*, const int64_t*, int64_t)>& exact_one)
            -> int {
        if (kind == KNHIP_IVF_PQ) {
            if (int rc = ensure_psum(idx)) return rc; // (the token streams / the half codebook follow the form)
        } else {
            if (int rc = ensure_mscan_norms(idx)) return rc;
        }
        int qt = mscan_queries_per_unit(kind, false);
        // IVF-Flat: the filter pass on the bf16 matrix pipe (mfma_scan_bf16.hip: up to 128 queries per unit); the sample
        // pass stays on the fp32 kernel (its units hold one or two queries: bound by the rows it reads, not by the products)
        const bool flat_b = kind == KNHIP_IVF_FLAT && idx->flat_bf16 && mscan_flat_bf16_qt(ms_nstep) > 0;
        if (flat_b) {
            qt = mscan_flat_bf16_qt(ms_nstep);
        }
        const int qt0 = mscan_queries_per_unit(kind, true);
        const int64_t units_bound = round_up(npairs / qt + std::min<int64_t>(nlist, npairs) + 1, 8);
        const int64_t sample = mscan_sample_rows();
        // (a query samples at most `sample` rows of non-empty lists: at most that many pairs)
        const int64_t np0 = std::min<int64_t>(npairs, nq * std::min<int64_t>(nprobe, sample));
        const int64_t bound0 = round_up(np0 / qt0 + std::min<int64_t>(nlist, np0) + 1, 8);
        HIP_TRY(ws->ms_units.reserve((size_t)std::max(units_bound, bound0) * sizeof(KnItem)));
        HIP_TRY(ws->ms_unit_off.reserve((size_t)(nlist + 1) * sizeof(int64_t)));
        HIP_TRY(ws->ms_nunits.reserve(sizeof(int64_t) + 2 * sizeof(double)));
        HIP_TRY(ws->ms_cand.reserve((size_t)nq * ms_cap * sizeof(int64_t)));
        HIP_TRY(ws->ms_cand_pess.reserve((size_t)nq * ms_cap * sizeof(float)));
        if (kind == KNHIP_IVF_SQ8) {
            HIP_TRY(ws->ms_eps_max.reserve((size_t)nq * sizeof(uint32_t)));
            HIP_TRY(hipMemsetAsync(ws->ms_eps_max.p, 0, (size_t)nq * sizeof(uint32_t), s));
        }
        HIP_TRY(ws->ms_cand_cnt.reserve((size_t)(2 * nq + 4) * sizeof(int32_t))); // counters, flags, any-flag, guard counters
        HIP_TRY(ws->dump.reserve((size_t)nq * sample * sizeof(float)));
        HIP_TRY(ws->sel_keys.reserve((size_t)nq * k * sizeof(int64_t)));
        HIP_TRY(ws->sel_d.reserve((size_t)nq * k * sizeof(float)));
        HIP_TRY(ws->ghist.reserve((size_t)nq * 64 * sizeof(uint32_t)));
        HIP_TRY(ws->gmeta.reserve((size_t)nq * sizeof(uint2)));
        int32_t* cand_cnt = ws->ms_cand_cnt.as<int32_t>();
        int32_t* overflow = cand_cnt + nq;
        MScanArgs m{};
        m.rows = idx->rows.p;
        m.xnorm = idx->xnorm.as<float>();
        m.xnorm_max = idx->xnorm_max;
        m.list_blk_off = idx->d_list_blk_off.as<int64_t>();
        m.list_len = idx->d_list_len.as<int64_t>();
        m.list_row_off = idx->d_list_row_off.as<int64_t>();
        m.ids = idx->ids.as<int64_t>();
        m.trained = idx->sq_trained.as<float>();
        m.centroids = idx->centroids.as<float>();
        m.d = d;
        m.nchunk = ms_nchunk;
        m.nstep = ms_nstep;
        m.queries = d_q;
        m.qnorm = ws->qnorm.as<float>();
        m.coarse_dis = cdis_p;
        m.nq = nq;
        m.nslot = nprobe;
        m.units = ws->ms_units.as<KnItem>();
        m.pairs = wt.pairs;
        m.nunits_dev = ws->ms_nunits.as<int64_t>();
        m.gthr = ws->gthr.as<float>();
        // |approx - exact| <= eps_scale * magnitude: see mfma_scan.hip
        const float eps_fp32 = (kind == KNHIP_IVF_FLAT ? 16.0f : 32.0f) * (float)d * 5.9604645e-8f;
        // (split-bf16 products drop lo lo + r_q x + q r_x <= 3 * 2^-16 ||q|| ||x||: mfma_scan_bf16.hip)
        m.eps_scale = eps_fp32 + (flat_b ? 6.103515625e-5f : 0.f);
        m.bitset = d_bitset;
        m.bitset_nbits = nbits;
        m.cand_cnt = cand_cnt;
        m.cand = ws->ms_cand.as<int64_t>();
        m.cand_pess = ws->ms_cand_pess.as<float>();
        m.eps_max = kind == KNHIP_IVF_SQ8 ? ws->ms_eps_max.as<uint32_t>() : nullptr;
        m.cap = ms_cap;
        m.overflow = overflow;
        m.gthr_rw = ws->gthr.as<float>();
        m.k = k;
        if (idx->cand_hist) {
            m.ghist = ws->ghist.as<uint32_t>();
            m.gmeta = ws->gmeta.as<uint2>();
        }
        if (kind == KNHIP_IVF_PQ) {
            // (retry round: one-query units, up to one per pair)
            HIP_TRY(ws->pq_recs.reserve((size_t)std::max<int64_t>(std::max(units_bound, bound0), npairs) * sizeof(P8Rec)));
            HIP_TRY(ws->pq_ctr.reserve(8 * 16 * sizeof(int32_t)));
            m.list_blk_off = nullptr;
            m.pq_sblk_off_r = idx->d_list_blk_off_r.as<int64_t>();
            m.pq_psum = idx->psum.as<float>();
            m.pq_cb_t = idx->cb_t.as<float4>();
            m.pq_precomp_t = idx->precomp_t.as<float>();
            m.pq_codes = idx->codes_aos.as<uint8_t>();
            m.pq_lut_mode = !is_l2 ? PQ_LUT_IP : (idx->use_precomp ? PQ_LUT_PRECOMP : PQ_LUT_RESIDUAL);
            m.pq_recs = ws->pq_recs.as<P8Rec>();
            m.pq_ctr = ws->pq_ctr.as<int32_t>();
        }
        bool pq_i8 = false; // IVF-PQ: the integer form of the filter (chosen after the sample pass)
        bool pq_dec = false; // IVF-PQ: the decode form (pq_decode.hip)
        // which second form the guard weighs against the half tables: the decode form (default; its eps lies between the
        // half tables' and the int8 tables', its units hold 128 queries) or, when asked for, the int8 tables
        const bool pqd_ok = kind == KNHIP_IVF_PQ && pqd_supports(idx->desc.pq_m, d);
        const bool want_dec = kind == KNHIP_IVF_PQ && pqd_ok && (idx->pqf_form == 0 || idx->pqf_form == 3);
        // IVF-PQ: the pairs are grouped by list (work table: four small, latency-bound kernels, ~0.25 ms per 10^4 queries at
        // C3) on a side stream while this stream runs the sample pass; only the cut into units waits for the form.
        SideJoin sj{ws, s};
        WorkTable wside = wt; // the table the filter pass reads (its own buffers when it is built on the side stream)
        if (!wt1_lazy && !env.no_side_stream) {
            // IVF-Flat / IVF-SQ8: the sample pass reads the split table built above; the all-probes table of the filter pass
            // goes to a second set of buffers and is built beside the sample pass (0.35 ms per batch at C2, 1.5 ms at C5)
            HIP_TRY(ws->list_count2.reserve((size_t)2 * nlist * sizeof(int32_t)));
            HIP_TRY(ws->list_cursor2.reserve((size_t)2 * nlist * sizeof(int32_t)));
            HIP_TRY(ws->list_pair_off2.reserve((size_t)(2 * nlist + 1) * sizeof(int64_t)));
            HIP_TRY(ws->list_item_off2.reserve((size_t)(2 * nlist + 1) * sizeof(int64_t)));
            HIP_TRY(ws->pairs2.reserve((size_t)npairs * sizeof(KnPair)));
            HIP_TRY(ws->items2.reserve((size_t)items_bound * sizeof(KnItem)));
            HIP_TRY(ws->nitems2.reserve(sizeof(int64_t)));
            wside.list_count = ws->list_count2.as<int32_t>();
            wside.list_cursor = ws->list_cursor2.as<int32_t>();
            wside.list_pair_off = ws->list_pair_off2.as<int64_t>();
            wside.list_item_off = ws->list_item_off2.as<int64_t>();
            wside.pairs = ws->pairs2.as<KnPair>();
            wside.items = ws->items2.as<KnItem>();
            wside.nitems = ws->nitems2.as<int64_t>();
            wside.scan_bytes = reinterpret_cast<double*>(ws->ms_nunits.as<int64_t>() + 1); // (bytes were counted above)
            if (ws->side == nullptr) {
                HIP_TRY(hipStreamCreateWithFlags(&ws->side, hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&ws->ev_fork, hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&ws->ev_join, hipEventDisableTiming));
            }
            HIP_TRY(hipEventRecord(ws->ev_fork, s));
            HIP_TRY(hipStreamWaitEvent(ws->side, ws->ev_fork, 0));
            HIP_TRY(launch_build_worktable(keys_p, nq, nprobe, nlist, qg, qg, idx->d_list_len.as<int64_t>(),
                                           idx->code_size, wside, ws->side, /*rank0_slot=*/-1));
            HIP_TRY(hipEventRecord(ws->ev_join, ws->side));
            sj.forked = true;
        }
        if (wt1_lazy && !env.no_side_stream) {
            if (ws->side == nullptr) {
                HIP_TRY(hipStreamCreateWithFlags(&ws->side, hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&ws->ev_fork, hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&ws->ev_join, hipEventDisableTiming));
            }
            HIP_TRY(hipEventRecord(ws->ev_fork, s));
            HIP_TRY(hipStreamWaitEvent(ws->side, ws->ev_fork, 0));
            WorkTable w2 = wt;
            HIP_TRY(launch_build_worktable(keys_p, nq, nprobe, nlist, qg, qg, idx->d_list_len.as<int64_t>(),
                                           idx->code_size, w2, ws->side, /*rank0_slot=*/-1));
            HIP_TRY(hipEventRecord(ws->ev_join, ws->side));
            sj.forked = true;
        }
        const bool wt2_done = sj.forked;
        auto launch_filter = [&](const MScanArgs& x, int64_t bound) -> hipError_t {
            return kind == KNHIP_IVF_FLAT ? ((flat_b && x.dump == nullptr) ? launch_mscan_flat_bf16(x, is_l2, bound, s)
                                                                            : launch_mscan_flat(x, is_l2, bound, s))
                 : kind == KNHIP_IVF_SQ8  ? launch_mscan_sq8(x, is_l2, bound, s)
                 : (pq_dec && x.dump == nullptr) ? launch_pqd(x, is_l2, bound, s)
                 : (pq_i8 && x.dump == nullptr)  ? launch_pqi(x, is_l2, bound, s)
                                                 : launch_pqf(x, is_l2, bound, s);
        };
        idx->rank0_phase_used = false;
        idx->last_pq_form = 0;
        {
            // phase 1: tau_q from a sample of the closest list (units of the rank-0 virtual lists [0, nlist), DUMP mode)
            StageTimer t(idx, s, KNHIP_STAGE_SCAN_RANK0);
            HIP_TRY(hipMemsetAsync(cand_cnt, 0, (size_t)(2 * nq + 1) * sizeof(int32_t), s));
            const bool want_i8 = kind == KNHIP_IVF_PQ && idx->pqf_form == 2;
            if (kind == KNHIP_IVF_PQ) {
                // (the sample pass below computes the queries' table statistics; the tables themselves follow the guard)
                HIP_TRY(ws->ms_qs.reserve((size_t)nq * 4 * sizeof(float)));
                HIP_TRY(ws->ms_nrow.reserve((size_t)nq * sizeof(int32_t)));
                m.pq_qs = ws->ms_qs.as<float>();
                if (want_i8) {
                    HIP_TRY(ws->ms_qi.reserve((size_t)nq * 256 * 32));
                    HIP_TRY(ws->ms_qis.reserve(((size_t)nq * 4 + 4) * sizeof(float))); // (+ the batch record)
                    HIP_TRY(ws->ms_qmu.reserve((size_t)nq * 32 * sizeof(float)));
                }
            } else if (kind == KNHIP_IVF_FLAT) {
                HIP_TRY(ws->qnorm.reserve((size_t)nq * sizeof(float)));
                m.qnorm = ws->qnorm.as<float>();
                HIP_TRY(launch_row_norms(d_q, nq, d, ws->qnorm.as<float>(), s));
            } else if (!is_l2) {
                // inner product: the query operand (scaled, split into two halves) is the same for every list
                const int ldq = ms_nstep * 32;
                HIP_TRY(ws->ms_qh.reserve((size_t)nq * ldq * 2));
                HIP_TRY(ws->ms_ql.reserve((size_t)nq * ldq * 2));
                HIP_TRY(ws->ms_qs.reserve((size_t)nq * 8 * sizeof(float)));
                HIP_TRY(launch_ms_sq8_query_prep(d_q, nq, d, ldq, idx->sq_trained.as<float>(), ws->ms_qh.p, ws->ms_ql.p,
                                                 ws->ms_qs.as<float>(), s));
                m.qh = ws->ms_qh.p;
                m.ql = ws->ms_ql.p;
                m.qs = ws->ms_qs.as<float>();
            }
            HIP_TRY(hipMemsetAsync(ws->ghist.p, 0, (size_t)nq * 64 * sizeof(uint32_t), s));
            MScanArgs ds = m;
            ds.eps_scale = eps_fp32; // (the sample pass runs the fp32 kernel)
            ds.dump = ws->dump.as<float>();
            ds.dump_stride = sample;
            ds.sample_cap = ms_sample_cap;
            ds.ghist = nullptr;
            if (kind == KNHIP_IVF_PQ) {
                // one workgroup per query: plan, fp32 table, sampled rows, and the statistics of both table forms
                int scap = (int)sample;
                if (env.pq_sample_rows > 0) { // (experiments: rows of the sample, at most)
                    scap = std::max(64, std::min((int)sample, env.pq_sample_rows));
                }
                HIP_TRY(launch_pq_sample(ds, keys_p, idx->cb.as<float4>(), nlist, std::max(1024, 8 * k), scap,
                                         idx->pabs_max, is_l2, ws->ms_nrow.as<int32_t>(), ws->ms_qs.as<float>(),
                                         want_i8 ? ws->ms_qis.as<float>() : nullptr,
                                         want_i8 ? ws->ms_qmu.as<float>() : nullptr, s, ws->gthr.as<float>(),
                                         ws->gmeta.as<uint2>(), k)); // (tau_q and the histogram range come out of it too)
            } else {
                HIP_TRY(launch_ms_units(wt.list_count, wt.list_pair_off, nlist, qt0, ws->ms_unit_off.as<int64_t>(),
                                        ws->ms_nunits.as<int64_t>(), ws->ms_units.as<KnItem>(),
                                        idx->d_list_len.as<int64_t>(), idx->code_size, nullptr, s));
                ds.sample_off = ws->ms_sample_off.as<int32_t>();
                HIP_TRY(launch_filter(ds, bound0));
            }
            if (kind != KNHIP_IVF_PQ) {
                HIP_TRY(launch_row_select_var(ws->dump.as<float>(), sample, keys_p, nprobe, idx->d_list_len.as<int64_t>(),
                                              nq, k, is_l2, ws->sel_keys.as<int64_t>(), ws->sel_d.as<float>(), s, sample,
                                              ws->ms_nrow.as<int32_t>()));
                HIP_TRY(launch_ms_tau(ws->sel_d.as<float>(), nq, k, is_l2, ws->gthr.as<float>(), ws->gmeta.as<uint2>(), s));
            }
        }
        {
            StageTimer t(idx, s, KNHIP_STAGE_TABLES); // (IVF_PQ: the filter's query operands / tables + the selectivity guard)
            const bool want_i8 = kind == KNHIP_IVF_PQ && idx->pqf_form == 2;
            if (kind == KNHIP_IVF_PQ) {
                // Three forms of the filter (pq_filter.hip, pq_decode.hip).  DECODE form: rows decoded once per (list, <= 128
                // queries), dense f16 contraction -- eps ~ 2^-9 B_q.  HALF tables: 8 queries per unit, eps = 2^-11 A_q: the
                // tightest.  INT8 tables: 16 queries per unit, eps 8 .. 20 x the half form's (kept for KNHIP_PQF_FORM=int8).
                // Selectivity guard: the sample dump predicts each query's candidate count under the eps of the half form and
                // of the second form (decode, or int8 when asked for); the batch takes the second form when that count is
                // small, else the half form, else -- data where even that lets a few percent of the rows through, so that the
                // exact finish would cost more than the exact scan -- the exact 4-query kernel.
                if (want_i8) { // (pass 1 -- ranges, midranges -- was part of the sample pass)
                    HIP_TRY(launch_pqi_query_table(d_q, idx->cb.as<float4>(), d, nq, is_l2, idx->pabs_max, ws->ms_qi.p,
                                                   ws->ms_qis.as<float>(), ws->ms_qmu.as<float>(), /*stats_done=*/true,
                                                   s));
                }
                if (want_dec) { // the queries as halves + their error records (cheap: the guard reads the records; on the side
                                // stream beside the sample pass it only shares the CUs with it: measured, no gain)
                    if (int rc = ensure_pqd(idx)) return rc;
                    HIP_TRY(ws->ms_qh16.reserve((size_t)nq * 128 * 2));
                    HIP_TRY(ws->ms_qd.reserve((size_t)nq * 4 * sizeof(float)));
                    HIP_TRY(launch_pqd_query_prep(d_q, idx->cb.as<float4>(), idx->pqd_st.as<float>(), nq, is_l2, idx->pabs_max,
                                                  ws->ms_qh16.p, ws->ms_qd.as<float>(), s));
                }
                const bool want2 = want_i8 || want_dec;
                const int form2 = want_dec ? 3 : 2;
                const float* qs2 = want_dec ? ws->ms_qd.as<float>() : want_i8 ? ws->ms_qis.as<float>() : nullptr;
                auto decide = [&](const int32_t* poor_h, int64_t n) -> int {
                    if (want2 && (idx->pqf_form == form2 || (int64_t)poor_h[1] * 4 <= n)) {
                        return form2;
                    }
                    if ((int64_t)poor_h[0] * 4 > n) {
                        return 0;
                    }
                    return 1;
                };
                int form = want2 ? form2 : 1; // (guard off: the form asked for)
                if (idx->pqf_guard) {
                    int32_t* poor = ws->ms_cand_cnt.as<int32_t>() + 2 * nq + 1;
                    // (the prediction pass -- 0.13 ms per 10^4 queries -- runs for the batches whose counters are looked at:
                    // the synchronous ones and every fourth of the others)
                    bool predicted = false;
                    auto predict = [&]() -> hipError_t {
                        if (predicted) {
                            return hipSuccess;
                        }
                        predicted = true;
                        return launch_pqf_predict(ws->dump.as<float>(), sample, ws->ms_nrow.as<int32_t>(), ws->gthr.as<float>(),
                                                  ws->ms_qs.as<float>(), qs2, keys_p, nprobe, nlist,
                                                  idx->d_list_len.as<int64_t>(), nq, ms_cap, k, is_l2, poor, s);
                    };
                    bool sync_now = true;
                    // (at most 64 (k, nprobe) pairs are remembered -- an entry owns a pinned buffer and an event; a caller
                    // that keeps inventing new pairs gets the synchronous decision)
                    bool cached = false;
                    {
                        std::lock_guard<std::mutex> lk(idx->mu);
                        cached = idx->guard_cache.size() < 64 || idx->guard_cache.count({k, nprobe}) != 0;
                    }
                    if (cached) {
                        std::lock_guard<std::mutex> lk(idx->mu);
                        knhip_index::GuardEntry& e = idx->guard_cache[{k, nprobe}];
                        if (e.pending && hipEventQuery(e.ev) == hipSuccess) { // the previous batch's counters are in
                            e.pending = false;
                            e.form = decide(e.h_poor, e.pending_nq);
                        }
                        e.age++;
                        const bool always_sync = env.guard_sync;
                        if (e.form > 0 && !always_sync && (e.age & 63) != 0) {
                            sync_now = false;
                            form = e.form;
                            if (!e.pending && (e.age & 3) == 0) {
                                HIP_TRY(predict());
                                if (e.h_poor == nullptr) {
                                    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e.h_poor), 2 * sizeof(int32_t)));
                                    HIP_TRY(hipEventCreateWithFlags(&e.ev, hipEventDisableTiming));
                                }
                                HIP_TRY(hipMemcpyAsync(e.h_poor, poor, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
                                HIP_TRY(hipEventRecord(e.ev, s));
                                e.pending = true;
                                e.pending_nq = nq;
                            }
                        }
                    }
                    if (sync_now) {
                        HIP_TRY(predict());
                        int32_t h_poor[2] = {0, 0};
                        HIP_TRY(hipMemcpyAsync(h_poor, poor, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
                        HIP_TRY(hipStreamSynchronize(s));
                        form = decide(h_poor, nq);
                        if (cached) {
                            std::lock_guard<std::mutex> lk(idx->mu);
                            knhip_index::GuardEntry& e = idx->guard_cache[{k, nprobe}];
                            if (!e.pending) {
                                e.form = form;
                            }
                        }
                    }
                    if (form == 0) {
                        return KNHIP_PQF_ABANDONED;
                    }
                }
                pq_i8 = form == 2;
                pq_dec = form == 3;
                idx->last_pq_form = form;
                if (form == 1) {
                    // the queries' half tables + scales (the same records the sample pass wrote)
                    if (int rc = ensure_pqf(idx)) return rc;
                    m.pq_codes_r = idx->rows_r.as<uint4>();
                    HIP_TRY(ws->ms_qh.reserve((size_t)nq * 256 * 32 * 2));
                    HIP_TRY(launch_pqf_query_table(d_q, idx->cb.as<float4>(), d, nq, is_l2, idx->pabs_max, ws->ms_qh.p,
                                                   ws->ms_qs.as<float>(), s));
                    m.pq_qh = ws->ms_qh.p;
                }
                if (pq_i8) {
                    if (int rc = ensure_pqi(idx)) return rc;
                    qt = 16;
                    HIP_TRY(ws->pq_recs16.reserve((size_t)std::max<int64_t>(std::max(units_bound, bound0), npairs) *
                                                  sizeof(P16Rec)));
                    m.pq_codes_r = idx->rows_r.as<uint4>();
                    m.pq_codes_i = idx->rows_i.as<uint4>();
                    m.pq_qi = ws->ms_qi.p;
                    m.pq_qis = ws->ms_qis.as<float>();
                    m.pq_recs16 = ws->pq_recs16.as<P16Rec>();
                }
                if (pq_dec) {
                    qt = PD_QT;
                    m.pq_cb16 = idx->pqd_cb16.p;
                    m.pq_qh16 = ws->ms_qh16.p;
                    m.pq_qd = ws->ms_qd.as<float>();
                    m.pq_sc = idx->pqd_st.as<float>();
                    m.pq_psum_s = idx->psum_s.as<float>();
                    // where a wave parks passing lanes beyond its LDS region (192 records): a list that is the closest list of many
                    // queries of the batch at once passes thousands of rows (C3: up to 3300 in one unit)
                    m.pq_spill_cap = 4 * 8192; // (per workgroup: 8192 records per wave, 671 MB of scratch at 256 workgroups)
                    if (idx->pqd_spill_cap > 0) {
                        m.pq_spill_cap = idx->pqd_spill_cap;
                    }
                    m.pq_spill_wgs = 512;
                    {
                        int dev = 0, ncu = 0;
                        (void)hipGetDevice(&dev);
                        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && ncu > 0) {
                            m.pq_spill_wgs = ncu;
                        }
                    }
                    HIP_TRY(ws->pq_spill.reserve((size_t)m.pq_spill_wgs * m.pq_spill_cap * 80));
                    m.pq_spill = static_cast<unsigned char*>(ws->pq_spill.p);
                }
            }
        }
        {
            // all probes of a list together: the work table again without the rank-0 split, its pairs cut into units
            StageTimer t(idx, s, KNHIP_STAGE_GROUP);
            WorkTable w2 = wside;
            if (!wt1_lazy) {
                w2.scan_bytes = reinterpret_cast<double*>(ws->ms_nunits.as<int64_t>() + 1); // (bytes were counted above)
            }
            if (wt2_done) {
                if (int rc = sj.join()) return rc;
            } else {
                HIP_TRY(launch_build_worktable(keys_p, nq, nprobe, nlist, qg, qg, idx->d_list_len.as<int64_t>(),
                                               idx->code_size, w2, s, /*rank0_slot=*/-1));
            }
            m.pairs = w2.pairs;
            HIP_TRY(launch_ms_units(w2.list_count + nlist, w2.list_pair_off + nlist, nlist, qt,
                                    ws->ms_unit_off.as<int64_t>(), ws->ms_nunits.as<int64_t>(),
                                    ws->ms_units.as<KnItem>(), idx->d_list_len.as<int64_t>(), idx->code_size,
                                    idx->scan_bytes_dev.as<double>() + 2, s));
        }
        {
            // phase 2: every (query, list) pair on the matrix cores
            StageTimer t(idx, s, KNHIP_STAGE_SCAN);
            HIP_TRY(launch_filter(m, units_bound));
        }
        {
            // phase 3: exact distances of the candidates -> final top-k.  phase 4: overflowed queries.  First a RETRY
            // (the exact k-th of the candidates a query gathered before it overflowed is a tight bound: its pairs are
            // filtered once more as one-query units, then finished); whatever overflows again, or had no bound and no
            // candidates, goes through the exact kernels (one-query items) and the ordinary merge.
            StageTimer t(idx, s, KNHIP_STAGE_MERGE);
            unsigned long long* counters = idx->coarse_fail_dev.as<unsigned long long>() + 1;
            MScanArgs mf = m;
            if (pq_i8) {
                mf.pq_qs = m.pq_qis; // (the finish kernel's pruning reads eps_base at [q][2] of either)
                mf.pq_prune_mu = 1;  // (... and the integer form's emission eps carries |sum of the per-m offsets|)
            }
            if (pq_dec) {
                mf.pq_qs = m.pq_qd;  // (eps_base at [q][2], like the table forms' records)
            }
            HIP_TRY(launch_mscan_finish(mf, kind, is_l2, keys_p, cdis_p, nprobe, k, d_out_d, d_out_i, counters, 1, s));
            HIP_TRY(launch_ms_flag_pairs(overflow, 2, keys_p, nq, nprobe, nlist, idx->d_list_len.as<int64_t>(), k,
                                         ws->items.as<KnItem>(), wt.pairs, wt.nitems, nullptr, s));
            MScanArgs r = m;
            r.pairs = wt.pairs; // (ms_flag_pairs wrote the retried queries' one-pair units there)
            r.units = ws->items.as<KnItem>();
            r.nunits_dev = wt.nitems;
            r.unit_loop = 1;
            r.ghist = nullptr; // (the retried rows were counted once already: counting them again would fake k candidates)
            r.gmeta = nullptr;
            HIP_TRY(launch_filter(r, npairs));
            HIP_TRY(launch_mscan_finish(mf, kind, is_l2, keys_p, cdis_p, nprobe, k, d_out_d, d_out_i, counters, 2, s));
            HIP_TRY(launch_ms_flag_pairs(overflow, 1, keys_p, nq, nprobe, nlist, idx->d_list_len.as<int64_t>(), k,
                                         ws->items.as<KnItem>(), wt.pairs, wt.nitems, ws->partial_i.as<int64_t>(), s));
            if (int rc = exact_one(ws->items.as<KnItem>(), wt.pairs, wt.nitems, std::min<int64_t>(npairs, 4096))) {
                return rc;
            }
            HIP_TRY(launch_merge_partials(ws->partial_d.as<float>(), ws->partial_i.as<int64_t>(), nq, nprobe, k,
                                          (int64_t)nprobe * k, k, is_l2, d_out_d, d_out_i, s, overflow));
        }
        return KNHIP_OK;
    };

    if (kind == KNHIP_IVF_FLAT) {
        FlatScanArgs a{};
        a.rows = idx->rows.as<float4>();
        a.list_blk_off = idx->d_list_blk_off.as<int64_t>();
        a.list_len = idx->d_list_len.as<int64_t>();
        a.list_row_off = idx->d_list_row_off.as<int64_t>();
        a.ids = idx->ids.as<int64_t>();
        a.d = d;
        a.nchunk = (d + 3) / 4;
        a.queries = d_q;
        a.nq = nq;
        a.items = wt.items;
        a.pairs = wt.pairs;
        a.nitems_dev = wt.nitems;
        a.bitset = d_bitset;
        a.bitset_nbits = nbits;
        a.partial_d = ws->partial_d.as<float>();
        a.partial_i = ws->partial_i.as<int64_t>();
        a.gthr = ws->gthr.as<float>();
        a.nslot = nprobe;
        a.k = k;
        a.row_scale = idx->row_scale.as<float>();
        a.cos_mode = idx->cos_mode;
        if (use_ms) {
            return run_mscan([&](const KnItem* items, const KnPair* pairs, const int64_t* nitems, int64_t grid) -> int {
                FlatScanArgs b = a;
                b.items = items;
                b.pairs = pairs;
                b.nitems_dev = nitems;
                b.item_loop = 1;
                HIP_TRY(launch_flat_scan(b, is_l2, false, grid, s, /*qg_override=*/1));
                return KNHIP_OK;
            });
        }
        StageTimer t(idx, s, KNHIP_STAGE_SCAN);
        HIP_TRY(launch_flat_scan(a, is_l2, false, items_bound, s));
    } else if (kind == KNHIP_IVF_PQ) {
        const int M = idx->desc.pq_m;
        const int mode = !is_l2 ? PQ_LUT_IP : (idx->use_precomp ? PQ_LUT_PRECOMP : PQ_LUT_RESIDUAL);
        if (mode != PQ_LUT_RESIDUAL && !(pq_use_q4 && !pq_rank0)) { // (pq_scan_q4 computes its tables from the codebook)
            HIP_TRY(ws->t2t.reserve((size_t)nq * 256 * M * sizeof(float)));
            StageTimer t(idx, s, KNHIP_STAGE_LUT);
            HIP_TRY(launch_pq_query_table(d_q, idx->cb.as<float>(), d, M, nq, ws->t2t.as<float>(), s));
        }
        PqScanArgs a{};
        a.codes_skew = idx->rows.as<uint4>();
        a.list_sblk_off = idx->d_list_blk_off.as<int64_t>();
        a.list_len = idx->d_list_len.as<int64_t>();
        a.list_row_off = idx->d_list_row_off.as<int64_t>();
        a.ids = idx->ids.as<int64_t>();
        a.precomp_t = idx->precomp_t.as<float>();
        a.cb = idx->cb.as<float>();
        a.centroids = idx->centroids.as<float>();
        a.d = d;
        a.lut_mode = mode;
        a.queries = d_q;
        a.t2t = ws->t2t.as<float>();
        a.coarse_dis = cdis_p;
        a.items = wt.items;
        a.pairs = wt.pairs;
        a.nitems_dev = wt.nitems;
        a.bitset = d_bitset;
        a.bitset_nbits = nbits;
        a.partial_d = ws->partial_d.as<float>();
        a.partial_i = ws->partial_i.as<int64_t>();
        a.gthr = ws->gthr.as<float>();
        a.nslot = nprobe;
        a.k = k;
        if (use_ms) {
            // half-precision prefilter + exact finish (pq_filter.hip); the queries that overflow twice take the exact
            // 4-query kernel over one-pair items
            if (pq_q4_ok) {
                a.codes_skew = idx->rows2.as<uint4>();
                a.list_sblk_off = idx->d_list_blk_off2.as<int64_t>();
                a.cb_t = idx->cb_t.as<float4>();
            }
            const int rc_ms = run_mscan([&](const KnItem* items, const KnPair* pairs, const int64_t* nitems, int64_t) -> int {
                PqScanArgs b = a;
                b.items = items;
                b.pairs = pairs;
                b.nitems_dev = nitems;
                b.item_lo = nullptr;
                b.item_hi = nitems;
                if (!pq_q4_ok) {
                    // k > 128: the systolic kernel, one workgroup per one-pair item (normally none: the workgroups of
                    // the launch read the item count and return)
                    if (!idx->skew_ready) {
                        if (int rc = build_pq_skew(idx)) return rc;
                    }
                    b.codes_skew = idx->rows.as<uint4>();
                    b.list_sblk_off = idx->d_list_blk_off.as<int64_t>();
                    HIP_TRY(launch_pq_scan(b, is_l2, M, npairs, s));
                    return KNHIP_OK;
                }
                HIP_TRY(ws->recs4.reserve((size_t)npairs * sizeof(P4Rec)));
                HIP_TRY(ws->q4_ctr.reserve(8 * 16 * sizeof(int32_t)));
                b.recs4 = ws->recs4.as<P4Rec>();
                b.q4_ctr = ws->q4_ctr.as<int32_t>();
                HIP_TRY(launch_pq_scan_q4(b, is_l2, npairs, s));
                return KNHIP_OK;
            });
            if (rc_ms != KNHIP_PQF_ABANDONED) {
                return rc_ms;
            }
            if (int rc = build_wt1()) return rc;
            // (the guard found the batch poorly selective: the exact 4-query kernel over the work table built above -- both
            // classes of the sample split are ordinary items of 4 pairs; gthr holds the sample's bounds, which are valid)
        }
        if (pq_use_v2) {
            a.codes_skew = idx->rows2.as<uint4>();
            a.list_sblk_off = idx->d_list_blk_off2.as<int64_t>();
            a.item_hi = wt.nitems;
            const int64_t stride = round_up(std::max<int64_t>(idx->max_list_len, 64), 64);
            if (pq_rank0) {
                // phase A: the rank-0 probe of every query (work items of virtual lists [0, nlist) come
                // first: worktable.hip) in dump mode, then radix select -> partial slot 0 + thresholds
                HIP_TRY(ws->dump.reserve((size_t)nq * stride * sizeof(float)));
                HIP_TRY(ws->sel_keys.reserve((size_t)nq * k * sizeof(int64_t)));
                HIP_TRY(ws->sel_d.reserve((size_t)nq * k * sizeof(float)));
                HIP_TRY(ws->ghist.reserve((size_t)nq * 64 * sizeof(uint32_t)));
                HIP_TRY(ws->gmeta.reserve((size_t)nq * sizeof(uint2)));
                a.dump = ws->dump.as<float>();
                a.dump_stride = stride;
                a.item_lo = nullptr;
                a.item_hi = wt.list_item_off + nlist; // items of the rank-0 virtual lists
                const int64_t boundA = round_up(nq / qg_rank0 + std::min<int64_t>(nlist, nq) + 1, 8);
                {
                    StageTimer t(idx, s, KNHIP_STAGE_SCAN_RANK0);
                    HIP_TRY(launch_pq_scan_v2(a, is_l2, true, boundA, s));
                    HIP_TRY(launch_rank0_select(a.dump, stride, keys_p, nprobe,
                                                idx->d_list_len.as<int64_t>(), idx->d_list_row_off.as<int64_t>(),
                                                idx->ids.as<int64_t>(), nq, k, is_l2, a.partial_d, a.partial_i,
                                                a.gthr, ws->sel_keys.as<int64_t>(), ws->sel_d.as<float>(),
                                                idx->cand_hist ? ws->ghist.as<uint32_t>() : nullptr,
                                                ws->gmeta.as<uint2>(), s));
                }
                if (idx->cand_hist) {
                    a.ghist = ws->ghist.as<uint32_t>();
                    a.gmeta = ws->gmeta.as<uint2>();
                }
                idx->rank0_phase_used = true;
                // phase B: every other probe
                a.item_lo = wt.list_item_off + nlist;
                a.item_hi = wt.nitems;
            } else {
                idx->rank0_phase_used = false;
            }
            StageTimer t(idx, s, KNHIP_STAGE_SCAN);
            if (pq_use_q4) {
                HIP_TRY(ws->recs4.reserve((size_t)items_bound * sizeof(P4Rec)));
                HIP_TRY(ws->q4_ctr.reserve(8 * 16 * sizeof(int32_t)));
                a.recs4 = ws->recs4.as<P4Rec>();
                a.q4_ctr = ws->q4_ctr.as<int32_t>();
                a.cb_t = idx->cb_t.as<float4>();
                HIP_TRY(launch_pq_scan_q4(a, is_l2, items_bound, s));
            } else {
                HIP_TRY(launch_pq_scan_v2(a, is_l2, false, items_bound, s));
            }
        } else {
            idx->rank0_phase_used = false;
            if (!idx->skew_ready) {
                if (int rc = build_pq_skew(idx)) return rc;
                a.codes_skew = idx->rows.as<uint4>();
            }
            StageTimer t(idx, s, KNHIP_STAGE_SCAN);
            HIP_TRY(launch_pq_scan(a, is_l2, M, items_bound, s));
        }
    } else { // IVF_SQ8
        SqScanArgs a{};
        a.rows = idx->rows.as<uint4>();
        a.list_blk_off = idx->d_list_blk_off.as<int64_t>();
        a.list_len = idx->d_list_len.as<int64_t>();
        a.list_row_off = idx->d_list_row_off.as<int64_t>();
        a.ids = idx->ids.as<int64_t>();
        a.trained = idx->sq_trained.as<float>();
        a.centroids = idx->centroids.as<float>();
        a.d = d;
        a.nchunk16 = (d + 15) / 16;
        a.queries = d_q;
        a.coarse_dis = cdis_p;
        a.items = wt.items;
        a.pairs = wt.pairs;
        a.nitems_dev = wt.nitems;
        a.bitset = d_bitset;
        a.bitset_nbits = nbits;
        a.partial_d = ws->partial_d.as<float>();
        a.partial_i = ws->partial_i.as<int64_t>();
        a.gthr = ws->gthr.as<float>();
        a.nslot = nprobe;
        a.k = k;
        if (use_ms) {
            return run_mscan([&](const KnItem* items, const KnPair* pairs, const int64_t* nitems, int64_t grid) -> int {
                SqScanArgs b = a;
                b.items = items;
                b.pairs = pairs;
                b.nitems_dev = nitems;
                b.item_loop = 1;
                HIP_TRY(launch_sq_scan(b, is_l2, grid, s, /*qg_override=*/1));
                return KNHIP_OK;
            });
        }
        StageTimer t(idx, s, KNHIP_STAGE_SCAN);
        HIP_TRY(launch_sq_scan(a, is_l2, items_bound, s));
    }
    {
        StageTimer t(idx, s, KNHIP_STAGE_MERGE);
        HIP_TRY(launch_merge_partials(ws->partial_d.as<float>(), ws->partial_i.as<int64_t>(), nq, nprobe, k,
                                      (int64_t)nprobe * k, k, is_l2, d_out_d, d_out_i, s));
    }
    return KNHIP_OK;
}

int validate_search(const knhip_index* idx, int64_t nq, int32_t k, int32_t& nprobe) {
    if (nq < 0 || k <= 0) {
        return fail(KNHIP_ERR_INVALID_ARGS, "nq must be >= 0 and k > 0");
    }
    if (k > KN_MAX_K) {
        return fail(KNHIP_ERR_INVALID_ARGS, "k > 1024 is not supported");
    }
    const int kind = idx->desc.kind;
    if (kind == KNHIP_BRUTE_FORCE) {
        if (!idx->has_data) {
            return fail(KNHIP_ERR_EMPTY_INDEX, "brute-force index holds no vectors");
        }
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
    if (!idx->has_data) {
        return fail(KNHIP_ERR_EMPTY_INDEX, "inverted lists not set");
    }
    if (nprobe <= 0) {
        return fail(KNHIP_ERR_INVALID_ARGS, "nprobe must be > 0");
    }
    if (nprobe > idx->nlist) {
        nprobe = (int32_t)idx->nlist; // IndexIVF.cpp:321-322
    }
    if ((size_t)nprobe > row_select_max_k()) {
        return fail(KNHIP_ERR_NOT_IMPLEMENTED, "nprobe > 65536 is not supported");
    }
    return KNHIP_OK;
}

// how many queries per batch so the scratch stays within ~8 GiB
int64_t query_batch(const knhip_index* idx, int64_t nq, int k, int nprobe) {
    double per_q;
    if (idx->desc.kind == KNHIP_BRUTE_FORCE) {
        const int64_t nb = idx->ntotal;
        int64_t chunk_rows = std::max<int64_t>(1024, round_up((nb + 511) / 512, 64));
        const int64_t nchunks = (nb + chunk_rows - 1) / chunk_rows;
        per_q = (double)nchunks * k * 12.0;
    } else {
        per_q = (double)idx->nlist * 4.0 + (double)nprobe * (12.0 + 8.0 + (double)k * 12.0);
        if ((size_t)nprobe > row_select_lds_max_k()) {
            per_q += 16.0 * nprobe; // sort scratch of the row selection (next power of two of nprobe, 8 bytes each)
        }
        if ((idx->desc.kind == KNHIP_IVF_FLAT || idx->desc.kind == KNHIP_IVF_SQ8) && idx->mscan != 0) {
            per_q += 4.0 * mscan_sample_rows() + 8.0 * 32768.0; // sample dump + candidate list (mfma_scan.hip)
        }
        if (idx->desc.kind == KNHIP_IVF_PQ && idx->pqf != 0) {
            // sample dump + candidate list + half table + one-pair records of the fallbacks (pq_filter.hip)
            per_q += 4.0 * mscan_sample_rows() + 12.0 * 32768.0 + 16384.0 + 8192.0 + (double)nprobe * (256.0 + 96.0);
        }
        if (idx->desc.kind == KNHIP_IVF_PQ) {
            per_q += 256.0 * idx->desc.pq_m * 4.0;
            if (!pq_scan_supported_m(idx->desc.pq_m)) { // (pq_scan_any.hip: four partial lists per probe)
                per_q += (double)nprobe * (pq_scan_any_parts(k) - 1) * (double)k * 12.0;
            }
        }
    }
    const double budget = 8.0 * 1024 * 1024 * 1024;
    int64_t qb = (int64_t)(budget / std::max(per_q, 1.0));
    qb = std::max<int64_t>(qb, 8);
    qb = std::min<int64_t>(qb, 65536 * 16);
    return std::min(qb, std::max<int64_t>(nq, 1));
}

} // namespace knhip_host

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int knhip_device_memory(int32_t device, int64_t* free_bytes, int64_t* total_bytes) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
        return fail(KNHIP_ERR_INVALID_ARGS, "device_memory: no such device");
    }
    DeviceGuard g(device);
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return KNHIP_OK;
}

int knhip_abi_version(void) {
    return KNHIP_ABI_VERSION;
}

int knhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        return 0;
    }
    return n;
}

const char* knhip_last_error(void) {
    return g_last_error.c_str();
}

int knhip_index_create(const knhip_desc* desc, knhip_index** out) {
    if (!desc || !out) {
        return fail(KNHIP_ERR_INVALID_ARGS, "null argument");
    }
    *out = nullptr;
    if (desc->kind < KNHIP_BRUTE_FORCE || desc->kind > KNHIP_IVF_SQ8) {
        return fail(KNHIP_ERR_INVALID_ARGS, "unknown index kind");
    }
    if (desc->metric != KNHIP_L2 && desc->metric != KNHIP_IP) {
        return fail(KNHIP_ERR_INVALID_ARGS, "metric must be L2 or IP (cosine is normalised IP)");
    }
    if (desc->dim <= 0 || desc->dim > 32768) {
        return fail(KNHIP_ERR_INVALID_ARGS, "dim out of range");
    }
    if (desc->kind != KNHIP_BRUTE_FORCE && (desc->nlist <= 0 || desc->nlist > (1 << 24))) {
        return fail(KNHIP_ERR_INVALID_ARGS, "nlist out of range");
    }
    if (desc->kind == KNHIP_IVF_PQ) {
        if (desc->pq_nbits < 1 || desc->pq_nbits > 8) {
            return fail(KNHIP_ERR_NOT_IMPLEMENTED, "PQ codes of 1 .. 8 bits are supported");
        }
        if (desc->pq_m <= 0 || desc->dim % desc->pq_m != 0) {
            return fail(KNHIP_ERR_INVALID_ARGS, "pq_m must divide dim");
        }
        // (8, 16, 32, 64: the fast kernels; every other width: pq_scan_any.hip)
        if (!pq_scan_supported_m(desc->pq_m) && !pq_scan_any_supports(desc->pq_m, desc->dim)) {
            return fail(KNHIP_ERR_NOT_IMPLEMENTED, "pq_m above 128, or sub-vectors of more than 144 dimensions");
        }
    }
    int ndev = knhip_device_count();
    if (ndev <= 0) {
        return fail(KNHIP_ERR_HIP_RUNTIME, "no HIP device visible");
    }
    if (desc->device < 0 || desc->device >= ndev) {
        return fail(KNHIP_ERR_INVALID_ARGS, "device ordinal out of range");
    }
    std::unique_ptr<knhip_index> idx(new knhip_index());
    idx->desc = *desc;
    idx->is_l2 = desc->metric == KNHIP_L2;
    idx->d = desc->dim;
    idx->nlist = desc->kind == KNHIP_BRUTE_FORCE ? 0 : desc->nlist;
    switch (desc->kind) {
        case KNHIP_BRUTE_FORCE:
        case KNHIP_IVF_FLAT:
            idx->code_size = (int64_t)desc->dim * 4;
            break;
        case KNHIP_IVF_PQ:
            idx->code_size = desc->pq_m; // (on the device: one byte per sub-quantizer whatever the width)
            idx->ksub = 1 << desc->pq_nbits;
            break;
        default:
            idx->code_size = desc->dim;
    }
    DeviceGuard g(desc->device);
    HIP_TRY(idx->scan_bytes_dev.alloc(3 * sizeof(double)));
    HIP_TRY(hipMemset(idx->scan_bytes_dev.p, 0, 3 * sizeof(double)));
    HIP_TRY(idx->coarse_fail_dev.alloc(8 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(idx->coarse_fail_dev.p, 0, 8 * sizeof(unsigned long long)));
    *out = idx.release();
    return KNHIP_OK;
}

void knhip_index_destroy(knhip_index* idx) {
    if (!idx) {
        return;
    }
    DeviceGuard g(idx->desc.device);
    (void)hipDeviceSynchronize();
    for (auto& p : idx->pending) {
        (void)hipEventDestroy(p.e0);
        (void)hipEventDestroy(p.e1);
    }
    for (auto& kv : idx->guard_cache) {
        if (kv.second.ev) (void)hipEventDestroy(kv.second.ev);
        if (kv.second.h_poor) (void)hipHostFree(kv.second.h_poor);
    }
    delete idx;
}

int knhip_index_set_coarse(knhip_index* idx, const float* centroids) {
    if (int rc = check_index(idx)) return rc;
    if (!centroids || idx->desc.kind == KNHIP_BRUTE_FORCE) {
        return fail(KNHIP_ERR_INVALID_ARGS, "set_coarse: bad arguments");
    }
    DeviceGuard g(idx->desc.device);
    if (int rc = upload(idx->centroids, centroids, (size_t)idx->nlist * idx->d * sizeof(float))) return rc;
    if (int rc = build_coarse_layout(idx)) return rc;
    idx->has_coarse = true;
    return maybe_build_precomp(idx);
}

int knhip_index_set_coarse_device(knhip_index* idx, const float* d_centroids) {
    if (int rc = check_index(idx)) return rc;
    if (!d_centroids || idx->desc.kind == KNHIP_BRUTE_FORCE) {
        return fail(KNHIP_ERR_INVALID_ARGS, "set_coarse_device: bad arguments");
    }
    DeviceGuard g(idx->desc.device);
    const size_t bytes = (size_t)idx->nlist * idx->d * sizeof(float);
    HIP_TRY(idx->centroids.alloc(bytes));
    HIP_TRY(hipMemcpy(idx->centroids.p, d_centroids, bytes, hipMemcpyDeviceToDevice));
    if (int rc = build_coarse_layout(idx)) return rc;
    idx->has_coarse = true;
    return maybe_build_precomp(idx);
}

// [M][ksub][dsub] -> [M][256][dsub]: the entries no code refers to repeat entry 0 (the encoder takes the FIRST minimum, table
// statistics see no value that is not a real entry's)
static std::vector<float> pad_codebook(const float* cb, int M, int ksub, int dsub) {
    std::vector<float> out((size_t)M * 256 * dsub);
    for (int m = 0; m < M; m++) {
        for (int c = 0; c < 256; c++) {
            const float* src = cb + ((size_t)m * ksub + (c < ksub ? c : 0)) * dsub;
            std::copy(src, src + dsub, out.begin() + ((size_t)m * 256 + c) * dsub);
        }
    }
    return out;
}

int knhip_index_set_pq(knhip_index* idx, const float* codebooks) {
    if (int rc = check_index(idx)) return rc;
    if (!codebooks || idx->desc.kind != KNHIP_IVF_PQ) {
        return fail(KNHIP_ERR_INVALID_ARGS, "set_pq: not an IVF_PQ index");
    }
    DeviceGuard g(idx->desc.device);
    if (idx->ksub == 256) {
        if (int rc = upload(idx->cb, codebooks, (size_t)256 * idx->d * sizeof(float))) return rc;
    } else {
        const std::vector<float> padded = pad_codebook(codebooks, idx->desc.pq_m, idx->ksub, idx->d / idx->desc.pq_m);
        if (int rc = upload(idx->cb, padded.data(), padded.size() * sizeof(float))) return rc;
    }
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

int knhip_index_set_sq(knhip_index* idx, const float* vmin, const float* vdiff) {
    if (int rc = check_index(idx)) return rc;
    if (!vmin || !vdiff || idx->desc.kind != KNHIP_IVF_SQ8) {
        return fail(KNHIP_ERR_INVALID_ARGS, "set_sq: not an IVF_SQ8 index");
    }
    DeviceGuard g(idx->desc.device);
    std::vector<float> t(2 * (size_t)idx->d);
    std::memcpy(t.data(), vmin, sizeof(float) * idx->d);
    std::memcpy(t.data() + idx->d, vdiff, sizeof(float) * idx->d);
    if (int rc = upload(idx->sq_trained, t.data(), t.size() * sizeof(float))) return rc;
    idx->has_sq = true;
    idx->xnorm_ready = false;
    return KNHIP_OK;
}

int knhip_index_set_row_scale(knhip_index* idx, const float* scale, int32_t mode) {
    if (int rc = check_index(idx)) return rc;
    const int kind = idx->desc.kind;
    if (!scale || mode == 0) {
        idx->row_scale.release();
        idx->cos_mode = 0;
        return KNHIP_OK;
    }
    if ((kind != KNHIP_BRUTE_FORCE && kind != KNHIP_IVF_FLAT) || idx->desc.metric != KNHIP_IP || (mode != 1 && mode != 2)) {
        return fail(KNHIP_ERR_INVALID_ARGS, "set_row_scale: inner-product BRUTE_FORCE / IVF_FLAT indexes, mode 1 or 2");
    }
    if (!idx->has_data) {
        return fail(KNHIP_ERR_EMPTY_INDEX, "set_row_scale: the index holds no vectors");
    }
    DeviceGuard g(idx->desc.device);
    // canonical entry order -> stored row positions (lists start at 64-row block boundaries); padding rows get 1
    std::vector<float> pos;
    if (kind == KNHIP_BRUTE_FORCE) {
        pos.assign((size_t)round_up(idx->ntotal, 64) + 2048, 1.0f);
        std::memcpy(pos.data(), scale, (size_t)idx->ntotal * sizeof(float));
    } else {
        pos.assign((size_t)idx->total_blk * 64 + 64, 1.0f);
        for (int64_t l = 0; l < idx->nlist; l++) {
            const int64_t len = idx->h_list_len[(size_t)l];
            if (len > 0) {
                std::memcpy(pos.data() + idx->h_list_blk_off[(size_t)l] * 64, scale + idx->h_list_row_off[(size_t)l],
                            (size_t)len * sizeof(float));
            }
        }
    }
    if (int rc = upload(idx->row_scale, pos.data(), pos.size() * sizeof(float))) return rc;
    idx->cos_mode = mode;
    return KNHIP_OK;
}

int knhip_index_add_lists(knhip_index* idx, const int64_t* list_sizes, const uint8_t* const* codes,
                          const int64_t* const* ids) {
    if (int rc = check_index(idx)) return rc;
    if (!list_sizes || !codes || !ids || idx->desc.kind == KNHIP_BRUTE_FORCE) {
        return fail(KNHIP_ERR_INVALID_ARGS, "add_lists: bad arguments");
    }
    DeviceGuard g(idx->desc.device);
    const int64_t nlist = idx->nlist;
    const int64_t cs = idx->code_size;
    // IVF_PQ with codes narrower than 8 bits: the caller's lists hold the reference's code bytes -- M indices of nbits bits as
    // a little-endian bit string of (M nbits + 7) / 8 bytes (ProductQuantizer.cpp:69, PQEncoderGeneric); unpacked here to one
    // byte per sub-quantizer
    const int nb_pq = idx->desc.kind == KNHIP_IVF_PQ ? idx->desc.pq_nbits : 8;
    const int64_t cs_in = nb_pq == 8 ? cs : ((int64_t)idx->desc.pq_m * nb_pq + 7) / 8;
    auto take_code = [&](uint8_t* dst, const uint8_t* src) {
        if (nb_pq == 8) {
            std::memcpy(dst, src, (size_t)cs);
            return;
        }
        for (int m = 0; m < (int)cs; m++) {
            const size_t bit = (size_t)m * nb_pq;
            const uint32_t w = (uint32_t)src[bit >> 3] | ((bit >> 3) + 1 < (size_t)cs_in ? (uint32_t)src[(bit >> 3) + 1] << 8 : 0u);
            dst[m] = (uint8_t)((w >> (bit & 7)) & ((1u << nb_pq) - 1u));
        }
    };
    std::vector<int64_t> off(nlist + 1, 0);
    for (int64_t l = 0; l < nlist; l++) {
        if (list_sizes[l] < 0 || (list_sizes[l] > 0 && (!codes[l] || !ids[l]))) {
            return fail(KNHIP_ERR_INVALID_ARGS, "add_lists: negative size or null list pointer");
        }
        off[l + 1] = off[l] + list_sizes[l];
    }
    const int64_t ntotal = off[nlist];
    // concatenate on the host, each list sorted by id (ties inside the kernels are broken by
    // storage position, which must therefore be the id order)
    std::vector<uint8_t> hc((size_t)ntotal * cs);
    std::vector<int64_t> hi((size_t)ntotal);
    std::vector<int64_t> perm;
    for (int64_t l = 0; l < nlist; l++) {
        const int64_t n = list_sizes[l];
        if (n == 0) {
            continue;
        }
        bool sorted = true;
        for (int64_t j = 1; j < n; j++) {
            if (ids[l][j] < ids[l][j - 1]) {
                sorted = false;
                break;
            }
        }
        if (sorted && nb_pq == 8) {
            std::memcpy(hc.data() + (size_t)off[l] * cs, codes[l], (size_t)n * cs);
            std::memcpy(hi.data() + off[l], ids[l], (size_t)n * sizeof(int64_t));
        } else if (sorted) {
            for (int64_t j = 0; j < n; j++) {
                take_code(hc.data() + (size_t)(off[l] + j) * cs, codes[l] + (size_t)j * cs_in);
            }
            std::memcpy(hi.data() + off[l], ids[l], (size_t)n * sizeof(int64_t));
        } else {
            perm.resize(n);
            std::iota(perm.begin(), perm.end(), 0);
            const int64_t* lid = ids[l];
            std::stable_sort(perm.begin(), perm.end(), [lid](int64_t a, int64_t b) { return lid[a] < lid[b]; });
            for (int64_t j = 0; j < n; j++) {
                take_code(hc.data() + (size_t)(off[l] + j) * cs, codes[l] + (size_t)perm[j] * cs_in);
                hi[off[l] + j] = lid[perm[j]];
            }
        }
    }
    DevBuf dc, di;
    if (int rc = upload(dc, hc.data(), hc.size())) return rc;
    if (int rc = upload(di, hi.data(), hi.size() * sizeof(int64_t))) return rc;
    return build_list_layout(idx, off, dc.as<uint8_t>(), di.as<int64_t>());
}

int knhip_index_set_lists_device(knhip_index* idx, const int64_t* list_offsets, const uint8_t* d_codes,
                                 const int64_t* d_ids) {
    if (int rc = check_index(idx)) return rc;
    if (!list_offsets || idx->desc.kind == KNHIP_BRUTE_FORCE) {
        return fail(KNHIP_ERR_INVALID_ARGS, "set_lists_device: bad arguments");
    }
    DeviceGuard g(idx->desc.device);
    std::vector<int64_t> off(list_offsets, list_offsets + idx->nlist + 1);
    if (off[0] != 0) {
        return fail(KNHIP_ERR_INVALID_ARGS, "list_offsets[0] must be 0");
    }
    return build_list_layout(idx, off, d_codes, d_ids);
}

} // extern "C"

namespace knhip_host {
int add_vectors_common(knhip_index* idx, int64_t n, const float* d_x, const int64_t* /*d_ids*/,
                              int64_t id_offset) {
    const int nchunk = (idx->d + 3) / 4;
    const int64_t nblk = (n + 63) / 64;
    idx->row_scale.release();
    idx->cos_mode = 0;
    HIP_TRY(idx->rows.alloc((size_t)nblk * nchunk * 64 * sizeof(float4)));
    HIP_TRY(launch_interleave_rows(d_x, n, idx->d, idx->rows.as<float4>(), 0, nullptr));
    if (idx->codes_aos.p != d_x) { // the raw rows stay resident: further Adds append to them, GetVectorByIds reads them
        HIP_TRY(idx->codes_aos.alloc((size_t)std::max<int64_t>(n, 1) * idx->d * sizeof(float)));
        HIP_TRY(hipMemcpy(idx->codes_aos.p, d_x, (size_t)n * idx->d * sizeof(float), hipMemcpyDeviceToDevice));
    }
    HIP_TRY(hipDeviceSynchronize());
    idx->ntotal = n;
    idx->id_offset = id_offset;
    idx->has_data = n > 0;
    idx->rg_seg_nseg = -1; // (the segment table of range_segments belongs to the old row count)
    idx->bf_split_ready = false;
    idx->rows_bs.release();
    idx->bf_norm.release();
    {
        const EnvLayout env = env_layout(); // (KNHIP_COARSE: the switch of the stage whose machinery this is)
        idx->bf_mfma = !env.bf_exact && !env.coarse_given;
    }
    return KNHIP_OK;
}
} // namespace knhip_host

extern "C" {

int knhip_index_add_vectors(knhip_index* idx, int64_t n, const float* x, const int64_t* ids,
                            int64_t id_offset) {
    if (int rc = check_index(idx)) return rc;
    if (idx->desc.kind != KNHIP_BRUTE_FORCE || n < 0 || (n > 0 && !x)) {
        return fail(KNHIP_ERR_INVALID_ARGS, "add_vectors: bad arguments");
    }
    if (ids) {
        return fail(KNHIP_ERR_NOT_IMPLEMENTED,
                    "explicit ids on a brute-force index: use id_offset (ids are row + offset)");
    }
    DeviceGuard g(idx->desc.device);
    DevBuf dx;
    if (int rc = upload(dx, x, (size_t)n * idx->d * sizeof(float))) return rc;
    return add_vectors_common(idx, n, dx.as<float>(), nullptr, id_offset);
}

int knhip_index_add_vectors_device(knhip_index* idx, int64_t n, const float* d_x, const int64_t* d_ids,
                                   int64_t id_offset) {
    if (int rc = check_index(idx)) return rc;
    if (idx->desc.kind != KNHIP_BRUTE_FORCE || n < 0 || (n > 0 && !d_x) || d_ids) {
        return fail(KNHIP_ERR_INVALID_ARGS, "add_vectors_device: bad arguments");
    }
    DeviceGuard g(idx->desc.device);
    return add_vectors_common(idx, n, d_x, nullptr, id_offset);
}

int64_t knhip_index_count(const knhip_index* idx) {
    return idx ? idx->ntotal : 0;
}

int knhip_index_get_desc(const knhip_index* idx, knhip_desc* out) {
    if (int rc = check_index(idx)) return rc;
    if (!out) {
        return fail(KNHIP_ERR_INVALID_ARGS, "get_desc: null output");
    }
    *out = idx->desc;
    return KNHIP_OK;
}

int64_t knhip_index_device_bytes(const knhip_index* idx) {
    return idx ? idx->device_bytes() : 0;
}

int64_t knhip_index_last_range_ranks(const knhip_index* idx) {
    return idx ? idx->last_range_ranks : -1;
}

int knhip_index_uses_precomputed_table(const knhip_index* idx) {
    return idx ? idx->use_precomp : 0;
}

// (defined behind the range pass it reuses)

int knhip_search_device(const knhip_index* idx, const float* d_queries, int64_t nq, int32_t k,
                        int32_t nprobe, const uint8_t* d_bitset, int64_t bitset_nbits,
                        int64_t* d_out_ids, float* d_out_dist, void* stream) {
    if (int rc = check_index(idx)) return rc;
    if (int rc = validate_search(idx, nq, k, nprobe)) return rc;
    if (nq == 0) {
        return KNHIP_OK;
    }
    if (!d_queries || !d_out_ids || !d_out_dist) {
        return fail(KNHIP_ERR_INVALID_ARGS, "null query/output pointer");
    }
    DeviceGuard g(idx->desc.device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    Workspace* ws = acquire_ws(idx, stream, true);
    std::lock_guard<std::mutex> ws_lock(ws->mu);  // two host threads enqueueing on one stream share this scratch
    const int64_t qb = query_batch(idx, nq, k, nprobe);
    if (idx->desc.kind != KNHIP_BRUTE_FORCE) {
        idx->coarse_flops += 2.0 * (double)nq * (double)idx->nlist * (double)idx->d;
    }
    for (int64_t q0 = 0; q0 < nq; q0 += qb) {
        const int64_t n = std::min(qb, nq - q0);
        if (int rc = search_batch_ties(idx, ws, d_queries + q0 * idx->d, n, k, nprobe, d_bitset, bitset_nbits,
                                       d_out_ids + q0 * k, d_out_dist + q0 * k, s, nullptr, nullptr)) {
            return rc;
        }
    }
    return KNHIP_OK;
}

int knhip_search_preassigned_device(const knhip_index* idx, const float* d_queries, int64_t nq, int32_t k,
                                    int32_t nprobe, const int64_t* d_keys, const float* d_coarse_dis,
                                    const uint8_t* d_bitset, int64_t bitset_nbits, int64_t* d_out_ids,
                                    float* d_out_dist, void* stream) {
    if (int rc = check_index(idx)) return rc;
    if (idx->desc.kind == KNHIP_BRUTE_FORCE) {
        return fail(KNHIP_ERR_INVALID_ARGS, "search_preassigned needs an IVF index");
    }
    const int32_t nprobe_in = nprobe;
    if (int rc = validate_search(idx, nq, k, nprobe)) return rc;
    if (nprobe != nprobe_in) {
        return fail(KNHIP_ERR_INVALID_ARGS, "search_preassigned: nprobe must not exceed nlist (keys are [nq][nprobe])");
    }
    if (nq == 0) {
        return KNHIP_OK;
    }
    if (!d_queries || !d_out_ids || !d_out_dist || !d_keys || !d_coarse_dis) {
        return fail(KNHIP_ERR_INVALID_ARGS, "null query/assignment/output pointer");
    }
    DeviceGuard g(idx->desc.device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    Workspace* ws = acquire_ws(idx, stream, true);
    std::lock_guard<std::mutex> ws_lock(ws->mu);  // two host threads enqueueing on one stream share this scratch
    const int64_t qb = query_batch(idx, nq, k, nprobe);
    for (int64_t q0 = 0; q0 < nq; q0 += qb) {
        const int64_t n = std::min(qb, nq - q0);
        if (int rc = search_batch_ties(idx, ws, d_queries + q0 * idx->d, n, k, nprobe, d_bitset, bitset_nbits,
                                       d_out_ids + q0 * k, d_out_dist + q0 * k, s, d_keys + q0 * nprobe,
                                       d_coarse_dis + q0 * nprobe)) {
            return rc;
        }
    }
    return KNHIP_OK;
}

// host boundary of Search(): queries up, (search [+ exact re-rank against the raw rows of `raw`]), results down
// the rows the second stage of knhip_search_refine* reads: fp32 rows of a BRUTE_FORCE index, or a quantised knhip_rows store
struct RefineStore {
    const void* rows = nullptr;   // device: [n][d] fp32 / 16-bit / 8-bit codes
    int64_t n = 0, id0 = 0;
    int row_type = 0;             // 0 fp32, 1 fp16, 2 bf16, 3 sq8
    const float* sq = nullptr;    // device: vmin[d], vdiff[d] (sq8)
};

static int search_host_impl(const knhip_index* idx, const RefineStore* raw, const float* queries, int64_t nq, int32_t k,
                            int32_t k_base, int32_t nprobe, const uint8_t* bitset, int64_t bitset_nbits,
                            int64_t* out_ids, float* out_dist) {
    if (int rc = check_index(idx)) return rc;
    const int32_t ks = raw ? k_base : k; // what the index is searched for
    if (raw && (k <= 0 || k_base < k)) {
        return fail(KNHIP_ERR_INVALID_ARGS, "search_refine: need k_base >= k > 0");
    }
    if (int rc = validate_search(idx, nq, ks, nprobe)) return rc;
    if (nq == 0) {
        return KNHIP_OK;
    }
    if (!queries || !out_ids || !out_dist) {
        return fail(KNHIP_ERR_INVALID_ARGS, "null query/output pointer");
    }
    DeviceGuard g(idx->desc.device);
    hipStream_t s = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    Workspace* ws = acquire_ws(idx, nullptr, false);
    int rc = KNHIP_OK;
    auto run = [&]() -> int {
        const size_t qbytes = (size_t)nq * idx->d * sizeof(float);
        HIP_TRY(ws->h_queries.reserve(qbytes));
        HIP_TRY(ws->h_out_d.reserve((size_t)nq * ks * sizeof(float)));
        HIP_TRY(ws->h_out_i.reserve((size_t)nq * ks * sizeof(int64_t)));
        HIP_TRY(hipMemcpyAsync(ws->h_queries.p, queries, qbytes, hipMemcpyHostToDevice, s));
        const uint8_t* d_bitset = nullptr;
        if (bitset && bitset_nbits > 0) {
            const size_t bb = (size_t)((bitset_nbits + 7) / 8);
            HIP_TRY(ws->h_bitset.reserve(bb));
            HIP_TRY(hipMemcpyAsync(ws->h_bitset.p, bitset, bb, hipMemcpyHostToDevice, s));
            d_bitset = ws->h_bitset.as<uint8_t>();
        }
        const int64_t qb = query_batch(idx, nq, ks, nprobe);
        if (idx->desc.kind != KNHIP_BRUTE_FORCE) {
            std::lock_guard<std::mutex> lk(idx->mu);
            idx->coarse_flops += 2.0 * (double)nq * (double)idx->nlist * (double)idx->d;
        }
        for (int64_t q0 = 0; q0 < nq; q0 += qb) {
            const int64_t n = std::min(qb, nq - q0);
            if (int r = search_batch_ties(idx, ws, ws->h_queries.as<float>() + q0 * idx->d, n, ks, nprobe, d_bitset,
                                          bitset_nbits, ws->h_out_i.as<int64_t>() + q0 * ks,
                                          ws->h_out_d.as<float>() + q0 * ks, s, nullptr, nullptr)) {
                return r;
            }
        }
        const float* res_d = ws->h_out_d.as<float>();
        const int64_t* res_i = ws->h_out_i.as<int64_t>();
        if (raw) { // IndexRefine::search second stage, on the device-resident raw rows
            HIP_TRY(ws->h_ref_d.reserve((size_t)nq * k * sizeof(float)));
            HIP_TRY(ws->h_ref_i.reserve((size_t)nq * k * sizeof(int64_t)));
            StageTimer t(idx, s, KNHIP_STAGE_REFINE);
            HIP_TRY(launch_refine(static_cast<const float*>(raw->rows), raw->n, raw->id0, idx->d,
                                  ws->h_queries.as<float>(), nq, res_i, k_base, k, idx->is_l2, ws->h_ref_d.as<float>(),
                                  ws->h_ref_i.as<int64_t>(), s, raw->row_type, raw->sq));
            res_d = ws->h_ref_d.as<float>();
            res_i = ws->h_ref_i.as<int64_t>();
        }
        HIP_TRY(hipMemcpyAsync(out_dist, res_d, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(out_ids, res_i, (size_t)nq * k * sizeof(int64_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        return KNHIP_OK;
    };
    rc = run();
    if (rc != KNHIP_OK) {
        (void)hipStreamSynchronize(s);
    }
    release_ws(idx, ws);
    (void)hipStreamDestroy(s);
    return rc;
}

int knhip_search(const knhip_index* idx, const float* queries, int64_t nq, int32_t k, int32_t nprobe,
                 const uint8_t* bitset, int64_t bitset_nbits, int64_t* out_ids, float* out_dist) {
    return search_host_impl(idx, nullptr, queries, nq, k, k, nprobe, bitset, bitset_nbits, out_ids, out_dist);
}

int knhip_search_refine(const knhip_index* idx, const knhip_index* raw, const float* queries, int64_t nq, int32_t k,
                        int32_t k_base, int32_t nprobe, const uint8_t* bitset, int64_t bitset_nbits, int64_t* out_ids,
                        float* out_dist) {
    if (!raw) {
        return fail(KNHIP_ERR_INVALID_ARGS, "search_refine: null raw-vector index");
    }
    if (int rc = check_index(idx)) return rc;
    if (raw->desc.kind != KNHIP_BRUTE_FORCE || raw->d != idx->d || raw->desc.device != idx->desc.device || !raw->has_data) {
        return fail(KNHIP_ERR_INVALID_ARGS, "search_refine: `raw` must be a filled brute-force index of the same "
                                            "dimension on the same device");
    }
    RefineStore st;
    st.rows = raw->codes_aos.p;
    st.n = raw->ntotal;
    st.id0 = raw->id_offset;
    return search_host_impl(idx, &st, queries, nq, k, k_base, nprobe, bitset, bitset_nbits, out_ids, out_dist);
}

// ---- quantised refine store (Knowhere's refine_type = fp16 / bf16 / sq8: the refine index is a faiss::IndexScalarQuantizer,
// reference src/index/refine/refine_utils.cc:99-160) ---------------------------------------------------------------------
} // extern "C"


extern "C" {


int knhip_search_refine_rows(const knhip_index* idx, const knhip_rows* rows, const float* queries, int64_t nq, int32_t k,
                             int32_t k_base, int32_t nprobe, const uint8_t* bitset, int64_t bitset_nbits, int64_t* out_ids,
                             float* out_dist) {
    if (!rows) {
        return fail(KNHIP_ERR_INVALID_ARGS, "search_refine_rows: null row store");
    }
    if (int rc = check_index(idx)) return rc;
    if (rows->d != idx->d || rows->device != idx->desc.device || rows->n <= 0 || !rows->trained) {
        return fail(KNHIP_ERR_INVALID_ARGS, "search_refine_rows: a filled store of the same dimension on the same device");
    }
    RefineStore st;
    st.rows = rows->codes.p;
    st.n = rows->n;
    st.id0 = 0;
    st.row_type = rows->row_type;
    st.sq = rows->ranged() ? rows->sq.as<float>() : nullptr;
    return search_host_impl(idx, &st, queries, nq, k, k_base, nprobe, bitset, bitset_nbits, out_ids, out_dist);
}

// rows of a brute-force index by id (GetVectorByIds): ids are row + id_offset
// IVF-Flat: ids sorted once with the column of their row in the interleaved store (16 bytes per vector, against the
// 4 d bytes of a second copy of the raw rows the node used to keep for this call: ADVICE round 2)
static int ensure_idmap(const knhip_index* cidx) {
    knhip_index* idx = const_cast<knhip_index*>(cidx);
    std::lock_guard<std::mutex> lk(idx->mu);
    if (idx->idmap_ready) {
        return KNHIP_OK;
    }
    const int64_t n = idx->ntotal;
    if (n > 0) {
        DevBuf col_tmp, tmp;
        const size_t tb = idmap_sort_tmp_bytes(n);
        HIP_TRY(col_tmp.alloc((size_t)n * sizeof(int64_t)));
        HIP_TRY(tmp.alloc(tb));
        HIP_TRY(idx->idmap_ids.alloc((size_t)n * sizeof(int64_t)));
        HIP_TRY(idx->idmap_col.alloc((size_t)n * sizeof(int64_t)));
        HIP_TRY(launch_idmap_build(idx->ids.as<int64_t>(), idx->d_list_row_off.as<int64_t>(),
                                   idx->d_list_blk_off.as<int64_t>(), idx->nlist, n, col_tmp.as<int64_t>(),
                                   idx->idmap_ids.as<int64_t>(), idx->idmap_col.as<int64_t>(), tmp.p, tb, nullptr));
        HIP_TRY(hipDeviceSynchronize());
    }
    idx->idmap_ready = true;
    return KNHIP_OK;
}

int knhip_index_get_vectors(const knhip_index* idx, int64_t n, const int64_t* ids, float* out) {
    if (int rc = check_index(idx)) return rc;
    const int kind = idx->desc.kind;
    if ((kind != KNHIP_BRUTE_FORCE && kind != KNHIP_IVF_FLAT) || n < 0 || (n > 0 && (!ids || !out))) {
        return fail(KNHIP_ERR_INVALID_ARGS, "get_vectors: brute-force or IVF-Flat index, ids and output required");
    }
    if (n == 0) {
        return KNHIP_OK;
    }
    DeviceGuard g(idx->desc.device);
    if (kind == KNHIP_IVF_FLAT) {
        if (int rc = ensure_idmap(idx)) return rc;
        DevBuf dw, dout, dmiss;
        if (int rc = upload(dw, ids, (size_t)n * sizeof(int64_t))) return rc;
        HIP_TRY(dout.alloc((size_t)n * idx->d * sizeof(float)));
        HIP_TRY(dmiss.alloc(sizeof(int32_t)));
        HIP_TRY(hipMemset(dmiss.p, 0, sizeof(int32_t)));
        HIP_TRY(launch_idmap_gather(dw.as<int64_t>(), n, idx->idmap_ids.as<int64_t>(), idx->idmap_col.as<int64_t>(),
                                    idx->ntotal, idx->rows.as<float4>(), idx->d, dout.as<float>(), dmiss.as<int32_t>(),
                                    nullptr));
        int32_t miss = 0;
        HIP_TRY(hipMemcpy(&miss, dmiss.p, sizeof(int32_t), hipMemcpyDeviceToHost));
        if (miss != 0) {
            return fail(KNHIP_ERR_INVALID_ARGS, "get_vectors: id not in the index");
        }
        HIP_TRY(hipMemcpy(out, dout.p, (size_t)n * idx->d * sizeof(float), hipMemcpyDeviceToHost));
        return KNHIP_OK;
    }
    std::vector<int64_t> rows((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        rows[(size_t)i] = ids[i] - idx->id_offset;
        if (rows[(size_t)i] < 0 || rows[(size_t)i] >= idx->ntotal) {
            return fail(KNHIP_ERR_INVALID_ARGS, "get_vectors: id out of range");
        }
    }
    DevBuf dr, dout;
    if (int rc = upload(dr, rows.data(), rows.size() * sizeof(int64_t))) return rc;
    HIP_TRY(dout.alloc((size_t)n * idx->d * sizeof(float)));
    HIP_TRY(launch_gather_rows(idx->codes_aos.as<float>(), dr.as<int64_t>(), n, idx->d, dout.as<float>(), nullptr));
    HIP_TRY(hipMemcpy(out, dout.p, (size_t)n * idx->d * sizeof(float), hipMemcpyDeviceToHost));
    return KNHIP_OK;
}

int knhip_index_find_vectors(const knhip_index* idx, int64_t n, const int64_t* ids, float* out, uint8_t* found) {
    if (int rc = check_index(idx)) return rc;
    const int kind = idx->desc.kind;
    if ((kind != KNHIP_BRUTE_FORCE && kind != KNHIP_IVF_FLAT) || n < 0 || (n > 0 && (!ids || !out || !found))) {
        return fail(KNHIP_ERR_INVALID_ARGS, "find_vectors: brute-force or IVF-Flat index, ids, output and flags required");
    }
    if (n == 0) {
        return KNHIP_OK;
    }
    DeviceGuard g(idx->desc.device);
    const int d = idx->d;
    if (kind == KNHIP_BRUTE_FORCE) {
        // rows by id range: gather the ones that live here
        std::vector<int64_t> rows, where;
        for (int64_t i = 0; i < n; i++) {
            const int64_t r = ids[i] - idx->id_offset;
            found[i] = (r >= 0 && r < idx->ntotal) ? 1 : 0;
            if (found[i]) {
                rows.push_back(r);
                where.push_back(i);
            }
        }
        if (rows.empty()) {
            return KNHIP_OK;
        }
        DevBuf dr, dout;
        if (int rc = upload(dr, rows.data(), rows.size() * sizeof(int64_t))) return rc;
        HIP_TRY(dout.alloc(rows.size() * d * sizeof(float)));
        HIP_TRY(launch_gather_rows(idx->codes_aos.as<float>(), dr.as<int64_t>(), (int64_t)rows.size(), d, dout.as<float>(),
                                   nullptr));
        std::vector<float> tmp(rows.size() * (size_t)d);
        HIP_TRY(hipMemcpy(tmp.data(), dout.p, tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (size_t j = 0; j < rows.size(); j++) {
            std::memcpy(out + where[j] * d, tmp.data() + j * d, (size_t)d * sizeof(float));
        }
        return KNHIP_OK;
    }
    if (idx->ntotal == 0) {
        std::memset(found, 0, (size_t)n);
        return KNHIP_OK;
    }
    if (int rc = ensure_idmap(idx)) return rc;
    DevBuf dw, dout, dmiss, dfound;
    if (int rc = upload(dw, ids, (size_t)n * sizeof(int64_t))) return rc;
    HIP_TRY(dout.alloc((size_t)n * d * sizeof(float)));
    HIP_TRY(dmiss.alloc(sizeof(int32_t)));
    HIP_TRY(dfound.alloc((size_t)n));
    HIP_TRY(hipMemset(dmiss.p, 0, sizeof(int32_t)));
    HIP_TRY(hipMemset(dfound.p, 0, (size_t)n));
    HIP_TRY(launch_idmap_gather(dw.as<int64_t>(), n, idx->idmap_ids.as<int64_t>(), idx->idmap_col.as<int64_t>(), idx->ntotal,
                                idx->rows.as<float4>(), d, dout.as<float>(), dmiss.as<int32_t>(), nullptr,
                                dfound.as<uint8_t>()));
    std::vector<float> tmp((size_t)n * d);
    HIP_TRY(hipMemcpy(found, dfound.p, (size_t)n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(tmp.data(), dout.p, tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; i++) {
        if (found[i]) {
            std::memcpy(out + i * d, tmp.data() + (size_t)i * d, (size_t)d * sizeof(float));
        }
    }
    return KNHIP_OK;
}

// ---- sharded refine: distances where the rows are, ONE selection over all of them --------------------------------------------
int knhip_refine_distances_device(int32_t metric, int32_t dim, const float* d_base, int64_t nbase, int64_t id_base,
                                  const float* d_queries, int64_t nq, const int64_t* d_cand_ids, int32_t k_base,
                                  float* d_out_dist, void* stream) {
    if (dim <= 0 || nbase < 0 || nq < 0 || k_base <= 0 || k_base > KN_MAX_K || (nbase > 0 && !d_base) || !d_queries ||
        !d_cand_ids || !d_out_dist || (metric != KNHIP_L2 && metric != KNHIP_IP)) {
        return fail(KNHIP_ERR_INVALID_ARGS, "refine_distances: bad arguments");
    }
    HIP_TRY(launch_refine(d_base, nbase, id_base, dim, d_queries, nq, d_cand_ids, k_base, 1, metric == KNHIP_L2, nullptr,
                          nullptr, static_cast<hipStream_t>(stream), 0, nullptr, nullptr, d_out_dist));
    return KNHIP_OK;
}

int knhip_refine_rows_distances_device(int32_t metric, const knhip_rows* rows, int64_t id_base, const float* d_queries,
                                       int64_t nq, const int64_t* d_cand_ids, int32_t k_base, float* d_out_dist,
                                       void* stream) {
    if (!rows || !rows->trained || nq < 0 || k_base <= 0 || k_base > KN_MAX_K || !d_queries || !d_cand_ids || !d_out_dist ||
        (metric != KNHIP_L2 && metric != KNHIP_IP)) {
        return fail(KNHIP_ERR_INVALID_ARGS, "refine_rows_distances: bad arguments");
    }
    HIP_TRY(launch_refine(static_cast<const float*>(rows->codes.p), rows->n, id_base, rows->d, d_queries, nq, d_cand_ids, k_base,
                          1, metric == KNHIP_L2, nullptr, nullptr, static_cast<hipStream_t>(stream), rows->row_type,
                          rows->ranged() ? rows->sq.as<float>() : nullptr, nullptr, d_out_dist));
    return KNHIP_OK;
}

int knhip_refine_combine_device(int32_t nshards, int64_t n, const float* d_parts, float* d_out, void* stream) {
    if (nshards <= 0 || n < 0 || !d_parts || !d_out) {
        return fail(KNHIP_ERR_INVALID_ARGS, "refine_combine: bad arguments");
    }
    HIP_TRY(launch_refine_combine(d_parts, nshards, n, d_out, static_cast<hipStream_t>(stream)));
    return KNHIP_OK;
}

int knhip_refine_select_device(int32_t metric, int64_t nq, const int64_t* d_cand_ids, const float* d_dist, int32_t k_base,
                               int32_t k, float* d_out_dist, int64_t* d_out_ids, void* stream) {
    if (nq < 0 || k <= 0 || k > KN_MAX_K || k_base < k || !d_cand_ids || !d_dist || !d_out_dist || !d_out_ids ||
        (metric != KNHIP_L2 && metric != KNHIP_IP)) {
        return fail(KNHIP_ERR_INVALID_ARGS, "refine_select: bad arguments");
    }
    HIP_TRY(launch_refine(nullptr, 0, 0, 4, nullptr, nq, d_cand_ids, k_base, k, metric == KNHIP_L2, d_out_dist, d_out_ids,
                          static_cast<hipStream_t>(stream), 0, nullptr, d_dist, nullptr));
    return KNHIP_OK;
}

// host form of the selection (a CPU-side merge of sharded refine results): the closed form of IndexRefine's reorder_2_heaps
// (Heap.h:657: the re-scored candidates pass a heap with strict-improve admission IN CANDIDATE ORDER) -- candidates up to the
// first -1 label, slots marked "not here" skipped
int knhip_refine_select_host(int32_t metric, int64_t nq, const int64_t* cand_ids, const float* dist, int32_t k_base, int32_t k,
                             float* out_dist, int64_t* out_ids) {
    if (nq < 0 || k <= 0 || k_base < k || !cand_ids || !dist || !out_dist || !out_ids || (metric != KNHIP_L2 && metric != KNHIP_IP)) {
        return fail(KNHIP_ERR_INVALID_ARGS, "refine_select_host: bad arguments");
    }
    const bool l2 = metric == KNHIP_L2;
    auto before = [l2](const std::pair<float, int64_t>& a, const std::pair<float, int64_t>& b) {
        if (l2) {
            return a.first < b.first || (a.first == b.first && a.second < b.second);
        }
        return a.first > b.first || (a.first == b.first && a.second > b.second);
    };
    std::vector<std::pair<float, int64_t>> arr, pool;
    for (int64_t q = 0; q < nq; q++) {
        arr.clear();
        for (int c = 0; c < k_base; c++) {
            const int64_t id = cand_ids[q * k_base + c];
            if (id == -1) {
                break;
            }
            uint32_t bits;
            std::memcpy(&bits, &dist[q * k_base + c], 4);
            if (id >= 0 && bits != 0xffffffffu) {
                arr.emplace_back(dist[q * k_base + c], id);
            }
        }
        pool = arr;
        std::sort(pool.begin(), pool.end(), before);
        if ((int64_t)pool.size() > k) {
            const float v = pool[(size_t)k - 1].first;
            pool.clear();
            int seen = 0; // arrivals with distance <= v so far
            for (const auto& a : arr) {
                const bool qual = l2 ? a.first <= v : a.first >= v;
                if (!qual) {
                    continue;
                }
                if (a.first != v || seen < k) {
                    pool.push_back(a);
                }
                seen++;
            }
            std::sort(pool.begin(), pool.end(), before);
        }
        for (int j = 0; j < k; j++) {
            if ((size_t)j < pool.size()) {
                out_dist[q * k + j] = pool[(size_t)j].first;
                out_ids[q * k + j] = pool[(size_t)j].second;
            } else {
                out_dist[q * k + j] = l2 ? FLT_MAX : -FLT_MAX;
                out_ids[q * k + j] = -1;
            }
        }
    }
    return KNHIP_OK;
}

void knhip_free(void* p) {
    std::free(p);
}

int knhip_coarse_search_device(const knhip_index* idx, const float* d_queries, int64_t nq,
                               int32_t nprobe, int64_t* d_out_keys, float* d_out_dist, void* stream) {
    if (int rc = check_index(idx)) return rc;
    if (idx->desc.kind == KNHIP_BRUTE_FORCE || !idx->has_coarse) {
        return fail(KNHIP_ERR_NOT_TRAINED, "coarse centroids not set");
    }
    if (nq <= 0 || nprobe <= 0 || !d_queries || !d_out_keys || !d_out_dist) {
        return fail(KNHIP_ERR_INVALID_ARGS, "coarse_search: bad arguments");
    }
    if (nprobe > idx->nlist || (size_t)nprobe > row_select_max_k()) {
        return fail(KNHIP_ERR_INVALID_ARGS, "coarse_search: nprobe out of range");
    }
    DeviceGuard g(idx->desc.device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    Workspace* ws = acquire_ws(idx, stream, true);
    std::lock_guard<std::mutex> ws_lock(ws->mu);  // two host threads enqueueing on one stream share this scratch
    const int64_t qb = std::max<int64_t>(1, std::min<int64_t>(nq, (int64_t)((4.0 * 1024 * 1024 * 1024) / (idx->nlist * 4.0))));
    for (int64_t q0 = 0; q0 < nq; q0 += qb) {
        const int64_t n = std::min(qb, nq - q0);
        StageTimer t(idx, s, KNHIP_STAGE_COARSE); // (a query-sharded coarse stage shows up in the rank's profile)
        if (int rc = coarse_stage(idx, ws, d_queries + q0 * idx->d, n, nprobe, d_out_keys + q0 * nprobe,
                                  d_out_dist + q0 * nprobe, s)) {
            return rc;
        }
    }
    return KNHIP_OK;
}

int knhip_merge_topk_device(int32_t metric, int64_t nq, int32_t k, int32_t nshard,
                            const float* d_dist_parts, const int64_t* d_ids_parts, float* d_out_dist,
                            int64_t* d_out_ids, void* stream) {
    if (nq < 0 || k <= 0 || k > KN_MAX_K || nshard <= 0 || !d_dist_parts || !d_ids_parts || !d_out_dist ||
        !d_out_ids) {
        return fail(KNHIP_ERR_INVALID_ARGS, "merge_topk: bad arguments");
    }
    // parts are [nshard][nq][k]: list (q, shard) starts at q * k + shard * nq * k
    HIP_TRY(launch_merge_partials(d_dist_parts, d_ids_parts, nq, nshard, k, k, nq * (int64_t)k,
                                  metric == KNHIP_L2, d_out_dist, d_out_ids,
                                  static_cast<hipStream_t>(stream)));
    return KNHIP_OK;
}

int knhip_merge_topk_host(int32_t metric, int64_t nq, int32_t k, int32_t nshard, const float* dist_parts,
                          const int64_t* ids_parts, float* out_dist, int64_t* out_ids) {
    if (nq < 0 || k <= 0 || nshard <= 0 || !dist_parts || !ids_parts || !out_dist || !out_ids) {
        return fail(KNHIP_ERR_INVALID_ARGS, "merge_topk_host: bad arguments");
    }
    const bool l2 = metric == KNHIP_L2;
    std::vector<std::pair<float, int64_t>> c;
    c.reserve((size_t)nshard * k);
    for (int64_t q = 0; q < nq; q++) {
        c.clear();
        for (int sh = 0; sh < nshard; sh++) {
            const float* dp = dist_parts + ((size_t)sh * nq + q) * k;
            const int64_t* ip = ids_parts + ((size_t)sh * nq + q) * k;
            for (int j = 0; j < k; j++) {
                if (ip[j] >= 0) {
                    c.emplace_back(dp[j], ip[j]);
                }
            }
        }
        // canonical order: L2 (dist asc, id asc); IP (dist desc, id desc)
        std::sort(c.begin(), c.end(), [l2](const std::pair<float, int64_t>& a, const std::pair<float, int64_t>& b) {
            if (l2) {
                return a.first < b.first || (a.first == b.first && a.second < b.second);
            }
            return a.first > b.first || (a.first == b.first && a.second > b.second);
        });
        for (int j = 0; j < k; j++) {
            if ((size_t)j < c.size()) {
                out_dist[q * k + j] = c[j].first;
                out_ids[q * k + j] = c[j].second;
            } else {
                out_dist[q * k + j] = l2 ? FLT_MAX : -FLT_MAX;
                out_ids[q * k + j] = -1;
            }
        }
    }
    return KNHIP_OK;
}

int knhip_refine_device(int32_t metric, int32_t dim, const float* d_base, int64_t nbase, int64_t id_base,
                        const float* d_queries, int64_t nq, const int64_t* d_cand_ids, int32_t k_base,
                        int32_t k, float* d_out_dist, int64_t* d_out_ids, void* stream) {
    if (dim <= 0 || nbase < 0 || nq < 0 || k <= 0 || k > KN_MAX_K || k_base < k || (nbase > 0 && !d_base) || !d_queries ||
        !d_cand_ids || !d_out_dist || !d_out_ids || (metric != KNHIP_L2 && metric != KNHIP_IP)) {
        return fail(KNHIP_ERR_INVALID_ARGS, "refine: bad arguments");
    }
    HIP_TRY(launch_refine(d_base, nbase, id_base, dim, d_queries, nq, d_cand_ids, k_base, k, metric == KNHIP_L2,
                          d_out_dist, d_out_ids, static_cast<hipStream_t>(stream)));
    return KNHIP_OK;
}

int knhip_refine_rows_device(int32_t metric, const knhip_rows* rows, int64_t id_base, const float* d_queries, int64_t nq,
                             const int64_t* d_cand_ids, int32_t k_base, int32_t k, float* d_out_dist, int64_t* d_out_ids,
                             void* stream) {
    if (!rows || !rows->trained || rows->n <= 0 || nq < 0 || k <= 0 || k > KN_MAX_K || k_base < k || !d_queries ||
        !d_cand_ids || !d_out_dist || !d_out_ids || (metric != KNHIP_L2 && metric != KNHIP_IP)) {
        return fail(KNHIP_ERR_INVALID_ARGS, "refine_rows: bad arguments");
    }
    HIP_TRY(launch_refine(static_cast<const float*>(rows->codes.p), rows->n, id_base, rows->d, d_queries, nq, d_cand_ids, k_base,
                          k, metric == KNHIP_L2, d_out_dist, d_out_ids, static_cast<hipStream_t>(stream), rows->row_type,
                          rows->ranged() ? rows->sq.as<float>() : nullptr));
    return KNHIP_OK;
}

} // extern "C"

extern "C" {

int knhip_index_get_coarse(const knhip_index* idx, float* centroids) {
    if (int rc = check_index(idx)) return rc;
    if (!idx->has_coarse || !centroids) {
        return fail(KNHIP_ERR_NOT_TRAINED, "coarse centroids not set");
    }
    DeviceGuard g(idx->desc.device);
    HIP_TRY(hipMemcpy(centroids, idx->centroids.p, (size_t)idx->nlist * idx->d * sizeof(float), hipMemcpyDeviceToHost));
    return KNHIP_OK;
}

int knhip_index_get_pq(const knhip_index* idx, float* codebooks) {
    if (int rc = check_index(idx)) return rc;
    if (!idx->has_pq || !codebooks) {
        return fail(KNHIP_ERR_NOT_TRAINED, "PQ codebooks not set");
    }
    DeviceGuard g(idx->desc.device);
    if (idx->ksub == 256) {
        HIP_TRY(hipMemcpy(codebooks, idx->cb.p, (size_t)256 * idx->d * sizeof(float), hipMemcpyDeviceToHost));
        return KNHIP_OK;
    }
    std::vector<float> h((size_t)256 * idx->d);
    HIP_TRY(hipMemcpy(h.data(), idx->cb.p, h.size() * sizeof(float), hipMemcpyDeviceToHost));
    const int M = idx->desc.pq_m, dsub = idx->d / M;
    for (int m = 0; m < M; m++) { // [M][256][dsub] -> [M][ksub][dsub]
        std::copy(h.begin() + (size_t)m * 256 * dsub, h.begin() + ((size_t)m * 256 + idx->ksub) * dsub,
                  codebooks + (size_t)m * idx->ksub * dsub);
    }
    return KNHIP_OK;
}

int knhip_index_get_sq(const knhip_index* idx, float* vmin, float* vdiff) {
    if (int rc = check_index(idx)) return rc;
    if (!idx->has_sq || !vmin || !vdiff) {
        return fail(KNHIP_ERR_NOT_TRAINED, "SQ parameters not set");
    }
    DeviceGuard g(idx->desc.device);
    HIP_TRY(hipMemcpy(vmin, idx->sq_trained.p, (size_t)idx->d * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(vdiff, idx->sq_trained.as<float>() + idx->d, (size_t)idx->d * sizeof(float), hipMemcpyDeviceToHost));
    return KNHIP_OK;
}

int knhip_index_get_list_sizes(const knhip_index* idx, int64_t* sizes) {
    if (int rc = check_index(idx)) return rc;
    if (idx->desc.kind == KNHIP_BRUTE_FORCE || !sizes) {
        return fail(KNHIP_ERR_INVALID_ARGS, "get_list_sizes: not an IVF index");
    }
    for (int64_t l = 0; l < idx->nlist; l++) {
        sizes[l] = idx->has_data ? idx->h_list_len[l] : 0;
    }
    return KNHIP_OK;
}

int knhip_index_get_lists(const knhip_index* idx, uint8_t* codes, int64_t* ids) {
    if (int rc = check_index(idx)) return rc;
    if (!idx->has_data) {
        return KNHIP_OK;
    }
    DeviceGuard g(idx->desc.device);
    std::lock_guard<std::mutex> add_lk(const_cast<knhip_index*>(idx)->add_mu);  // (not while an Add rebuilds the lists)
    if (codes) {
        if (int rc = ensure_aos(idx)) return rc;
        const size_t b = idx->desc.kind == KNHIP_BRUTE_FORCE ? (size_t)idx->ntotal * idx->d * sizeof(float)
                                                             : (size_t)idx->ntotal * idx->code_size;
        if (idx->desc.kind == KNHIP_IVF_PQ && idx->desc.pq_nbits != 8) {
            // -> the reference's code bytes (see knhip_index_add_lists)
            const int nb_pq = idx->desc.pq_nbits, M = idx->desc.pq_m;
            const size_t cs_out = ((size_t)M * nb_pq + 7) / 8;
            std::vector<uint8_t> h(b);
            HIP_TRY(hipMemcpy(h.data(), idx->codes_aos.p, b, hipMemcpyDeviceToHost));
            std::memset(codes, 0, (size_t)idx->ntotal * cs_out);
            for (int64_t i = 0; i < idx->ntotal; i++) {
                uint8_t* dst = codes + (size_t)i * cs_out;
                const uint8_t* src = h.data() + (size_t)i * M;
                for (int m = 0; m < M; m++) {
                    const size_t bit = (size_t)m * nb_pq;
                    const uint32_t v = (uint32_t)src[m] << (bit & 7);
                    dst[bit >> 3] |= (uint8_t)v;
                    if ((bit & 7) + (size_t)nb_pq > 8) {
                        dst[(bit >> 3) + 1] |= (uint8_t)(v >> 8);
                    }
                }
            }
        } else {
            HIP_TRY(hipMemcpy(codes, idx->codes_aos.p, b, hipMemcpyDeviceToHost));
        }
    }
    if (ids && idx->desc.kind != KNHIP_BRUTE_FORCE) {
        HIP_TRY(hipMemcpy(ids, idx->ids.p, (size_t)idx->ntotal * sizeof(int64_t), hipMemcpyDeviceToHost));
    }
    return KNHIP_OK;
}

int knhip_index_get_vectors_device(const knhip_index* idx, const float** d_rows) {
    if (int rc = check_index(idx)) return rc;
    if (idx->desc.kind != KNHIP_BRUTE_FORCE || !d_rows) {
        return fail(KNHIP_ERR_INVALID_ARGS, "get_vectors_device: brute-force index only");
    }
    *d_rows = idx->codes_aos.as<float>();
    return KNHIP_OK;
}

int knhip_profile_enable(knhip_index* idx, int on) {
    if (int rc = check_index(idx)) return rc;
    idx->prof = on != 0;
    return KNHIP_OK;
}

static void drain_pending(const knhip_index* idx) {
    std::vector<PendingEvent> p;
    {
        std::lock_guard<std::mutex> lk(idx->mu);
        p.swap(idx->pending);
    }
    for (auto& e : p) {
        float ms = 0.f;
        if (hipEventSynchronize(e.e1) == hipSuccess && hipEventElapsedTime(&ms, e.e0, e.e1) == hipSuccess) {
            idx->times.ms[e.stage] += ms;
            idx->times.launches[e.stage] += 1;
        }
        (void)hipEventDestroy(e.e0);
        (void)hipEventDestroy(e.e1);
    }
}

int knhip_profile_reset(knhip_index* idx) {
    if (int rc = check_index(idx)) return rc;
    DeviceGuard g(idx->desc.device);
    drain_pending(idx);
    std::memset(&idx->times, 0, sizeof(idx->times));
    idx->coarse_flops = 0;
    idx->tie_queries = 0;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(idx->scan_bytes_dev.p, 0, 3 * sizeof(double)));
    HIP_TRY(hipMemset(idx->coarse_fail_dev.p, 0, 8 * sizeof(unsigned long long)));
    return KNHIP_OK;
}

int knhip_profile_get(const knhip_index* idx, knhip_stage_times* out) {
    if (int rc = check_index(idx)) return rc;
    if (!out) {
        return fail(KNHIP_ERR_INVALID_ARGS, "null output");
    }
    DeviceGuard g(idx->desc.device);
    HIP_TRY(hipDeviceSynchronize());
    drain_pending(idx);
    double sb[3] = {0, 0, 0};
    HIP_TRY(hipMemcpy(sb, idx->scan_bytes_dev.p, 3 * sizeof(double), hipMemcpyDeviceToHost));
    *out = idx->times;
    out->scan_bytes = sb[0];
    out->scan_bytes_rank0 = idx->rank0_phase_used ? sb[1] : 0.0;
    out->coarse_flops = idx->coarse_flops;
    out->tie_queries = idx->tie_queries;
    out->scan_items = idx->last_items_bound;
    // coarse certificate failures; prefilter paths: finished / overflowed queries, candidates, exact recomputations
    unsigned long long nf[6] = {0, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpy(nf, idx->coarse_fail_dev.p, sizeof(nf), hipMemcpyDeviceToHost));
    out->tie_anomalies = (int64_t)(nf[5] & 0xffffffffull);
    out->coarse_fallback_queries = (int64_t)nf[0];
    out->mscan_queries = (int64_t)nf[1];
    out->mscan_overflow_queries = (int64_t)nf[2];
    out->mscan_candidates = (int64_t)nf[3];
    out->mscan_recomputed = (int64_t)nf[4];
    out->pq_filter_form = idx->desc.kind == KNHIP_BRUTE_FORCE ? (idx->last_bf_mfma ? 10 : 0) : idx->last_pq_form;
    out->mscan_stream_bytes = sb[2];
    return KNHIP_OK;
}

const char* knhip_stage_kernel_name(int stage, int kind) {
    // (the kernels of a large batch on the prefilter paths; small batches and fallbacks take the exact kernels named last)
    switch (stage) {
        case KNHIP_STAGE_COARSE:
            return "coarse_bf16_kernel (two passes)+coarse_bound_kernel+coarse_rerank_kernel | coarse_gemm_kernel+row_select_thr_kernel";
        case KNHIP_STAGE_GROUP:
            return "wt_*_kernel+ms_unit*_kernel";
        case KNHIP_STAGE_LUT:
            return "pq_query_table_kernel";
        case KNHIP_STAGE_SCAN_RANK0:
            return kind == KNHIP_IVF_PQ ? "pq_sample_kernel | pq_scan_v2_kernel<DUMP>+row_select_kernel"
                                        : "mscan_flat_kernel<DUMP> | mscan_sq8_kernel<DUMP> +row_select_kernel+ms_tau_kernel";
        case KNHIP_STAGE_TABLES:
            return "pqi_query_table_kernel | pqf_query_table_kernel +pqf_predict_kernel";
        case KNHIP_STAGE_SCAN:
            return kind == KNHIP_IVF_PQ    ? "pqi_kernel | pqf_kernel | pq_scan_q4_kernel | pq_scan_kernel | pq_scan_any_kernel"
                   : kind == KNHIP_IVF_SQ8 ? "mscan_sq8_kernel | sq_scan_kernel"
                                           : "mscan_flatb_kernel | mscan_flat_kernel | flat_scan_kernel";
        case KNHIP_STAGE_MERGE:
            return "mscan_finish_kernel | merge_partials_kernel";
        case KNHIP_STAGE_REFINE:
            return "refine_kernel";
        case KNHIP_STAGE_TIES:
            return "tie_detect_kernel+range dump pass+tie_resolve_kernel";
        default:
            return "other";
    }
}

} // extern "C"
