// knowhere_amd/csrc/knhip_api_range.hip -- RangeSearch and the boundary-tie rule behind the C ABI (include/knhip.h): the dump pass
// over a search's probed lists (range_batch), rank-wave probing with the reference's early stop, the first-come admission
// rule at the k-th boundary for one index (search_batch_ties) and its pieces for a list-sharded one (knhip_tie_*).
#include "knhip_internal.h"

extern "C" {

// ---- range search ----------------------------------------------------------------------------------------
// Every probed list is scanned in dump mode (all distances -> dist[q][column]); range.hip then counts the
// hits per (query, probe rank), applies the reference's early stop and compacts the survivors in the
// reference's emission order.  One batch of queries (device pointers); results appended to host vectors.
// Tie resolution of Search() reuses this pass (search_batch_ties): d_radius_q = one (inclusive) radius per query,
// nprobe_limit = the search's nprobe (the lists of ranks [0, nprobe_limit) only, no early stop), pre_keys / pre_cdis =
// a given coarse assignment [nq][nprobe_limit] (search_preassigned) -- the hits then come out in the reference's SCAN
// order (probe rank, storage position), which is what its first-come admission depends on.
static int range_batch(const knhip_index* idx, Workspace* ws, const float* d_q, int64_t nq, float radius,
                       int max_empty, const uint8_t* d_bitset, int64_t nbits, const int64_t* d_seg /*3 x nseg*/,
                       int64_t nseg, int64_t ncol, int64_t* h_lims /*nq + 1, relative*/, std::vector<int64_t>& out_i,
                       std::vector<float>& out_d, hipStream_t s, const float* d_radius_q = nullptr,
                       int nprobe_limit = 0, const int64_t* pre_keys = nullptr, const float* pre_cdis = nullptr,
                       RangeArgs* dump_only_out = nullptr, std::vector<int32_t>* out_cnt = nullptr) {
    const int kind = idx->desc.kind;
    const bool is_l2 = idx->is_l2;
    const int d = idx->d;
    const int nprobe = (nprobe_limit > 0 && kind != KNHIP_BRUTE_FORCE) ? std::min<int64_t>(nprobe_limit, nseg) : (int)nseg;
    const bool all_lists = nprobe == (int)nseg;
    HIP_TRY(ws->dump.reserve((size_t)nq * ncol * sizeof(float)));
    RangeArgs r{};
    r.dist = ws->dump.as<float>();
    r.ncol = ncol;
    r.seg_col = d_seg;
    r.seg_idpos = d_seg + nseg;
    r.seg_len = d_seg + 2 * nseg;
    r.nprobe = nprobe;
    r.radius = radius;
    r.radius_q = d_radius_q;
    r.inclusive = d_radius_q != nullptr ? 1 : 0;
    r.bitset = d_bitset;
    r.bitset_nbits = nbits;
    FlatScanArgs fc{};
    if (kind == KNHIP_BRUTE_FORCE || kind == KNHIP_IVF_FLAT) {
        fc.rows = idx->rows.as<float4>();
        fc.nrows = ncol;
        fc.chunk_rows = std::max<int64_t>(1024, round_up((ncol + 1023) / 1024, 64));
        fc.d = d;
        fc.nchunk = (d + 3) / 4;
        fc.queries = d_q;
        fc.nq = nq;
        fc.row_scale = idx->row_scale.as<float>();
        fc.cos_mode = idx->cos_mode;
    }
    if (kind == KNHIP_BRUTE_FORCE) {
        HIP_TRY(launch_flat_full(fc, is_l2, ws->dump.as<float>(), nullptr, 0, nullptr, s));
        r.ids = nullptr;
        r.id_offset = idx->id_offset;
        r.order = nullptr;
        max_empty = 0; // IndexFlat::range_search has no early stop
    } else {
        HIP_TRY(ws->keys.reserve((size_t)nq * nprobe * sizeof(int64_t)));
        HIP_TRY(ws->cdis.reserve((size_t)nq * nprobe * sizeof(float)));
        if (pre_keys != nullptr) {
            HIP_TRY(hipMemcpyAsync(ws->keys.p, pre_keys, (size_t)nq * nprobe * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
            HIP_TRY(hipMemcpyAsync(ws->cdis.p, pre_cdis, (size_t)nq * nprobe * sizeof(float), hipMemcpyDeviceToDevice, s));
        } else if (int rc = coarse_stage(idx, ws, d_q, nq, nprobe, ws->keys.as<int64_t>(), ws->cdis.as<float>(), s)) {
            return rc;
        }
        r.ids = idx->ids.as<int64_t>();
        r.order = ws->keys.as<int64_t>();
    }
    // every distance of the lists keys_w[q][0 .. W) (-1: none) -> dump[q][column]
    auto scan_dump = [&](const int64_t* keys_w, const float* cdis_w, int W) -> int {
        if (kind == KNHIP_IVF_FLAT) {
            if (W == nprobe && all_lists) {
                // all lists of every query: the dense all-pairs kernel (rows shared by eight queries)
                HIP_TRY(launch_flat_full(fc, is_l2, ws->dump.as<float>(), nullptr, 0, nullptr, s));
            } else {
                HIP_TRY(launch_range_flat_dump(fc, keys_w, nq, W, idx->nlist, r.seg_col, r.seg_len, ws->dump.as<float>(),
                                               ncol, is_l2, s));
            }
            return KNHIP_OK;
        }
        const int64_t nlist = idx->nlist;
        const int64_t npairs = nq * W;
        if (kind == KNHIP_IVF_PQ && !(idx->pq_v2 && idx->desc.pq_m == 32)) {
            // code widths without a dump mode in the fast ADC kernels: the plain exact ADC kernel of range.hip
            const int M = idx->desc.pq_m;
            const int mode = !is_l2 ? PQ_LUT_IP : (idx->use_precomp ? PQ_LUT_PRECOMP : PQ_LUT_RESIDUAL);
            PqDumpArgs a{};
            a.dist = ws->dump.as<float>();
            a.ncol = ncol;
            a.keys = keys_w;
            a.coarse_dis = cdis_w;
            a.nprobe = W;
            a.nlist = nlist;
            a.list_len = idx->d_list_len.as<int64_t>();
            a.list_row_off = idx->d_list_row_off.as<int64_t>();
            a.codes = idx->codes_aos.as<uint8_t>();
            a.M = M;
            a.d = d;
            a.lut_mode = mode;
            a.t2t = ws->t2t.as<float>();
            a.precomp_t = idx->precomp_t.as<float>();
            a.cb = idx->cb.as<float>();
            a.centroids = idx->centroids.as<float>();
            a.queries = d_q;
            HIP_TRY(launch_pq_adc_dump(a, nq, is_l2, s));
            return KNHIP_OK;
        }
        const int qg = kind == KNHIP_IVF_PQ ? pq_scan_qg(idx->desc.pq_m) : 8;
        const int64_t items_bound = round_up(npairs / qg + std::min<int64_t>(2 * nlist, npairs) + 1, 8);
        HIP_TRY(ws->list_count.reserve((size_t)2 * nlist * sizeof(int32_t)));
        HIP_TRY(ws->list_cursor.reserve((size_t)2 * nlist * sizeof(int32_t)));
        HIP_TRY(ws->list_pair_off.reserve((size_t)(2 * nlist + 1) * sizeof(int64_t)));
        HIP_TRY(ws->list_item_off.reserve((size_t)(2 * nlist + 1) * sizeof(int64_t)));
        HIP_TRY(ws->pairs.reserve((size_t)npairs * sizeof(KnPair)));
        HIP_TRY(ws->items.reserve((size_t)items_bound * sizeof(KnItem)));
        HIP_TRY(ws->nitems.reserve(sizeof(int64_t)));
        HIP_TRY(ws->gthr.reserve((size_t)nq * sizeof(float)));
        HIP_TRY(launch_fill_f32(ws->gthr.as<float>(), nq, is_l2 ? FLT_MAX : -FLT_MAX, s));
        WorkTable wt{};
        wt.list_count = ws->list_count.as<int32_t>();
        wt.list_cursor = ws->list_cursor.as<int32_t>();
        wt.list_pair_off = ws->list_pair_off.as<int64_t>();
        wt.list_item_off = ws->list_item_off.as<int64_t>();
        wt.pairs = ws->pairs.as<KnPair>();
        wt.items = ws->items.as<KnItem>();
        wt.nitems = ws->nitems.as<int64_t>();
        wt.scan_bytes = idx->scan_bytes_dev.as<double>();
        if (npairs <= 2048 && npairs <= items_bound) {
            // (one or two queries -- the boundary rule's flagged ones --: one item per pair instead of the grouped table)
            HIP_TRY(launch_direct_items(keys_w, nq, W, nlist, idx->d_list_len.as<int64_t>(), wt, s));
        } else {
            HIP_TRY(launch_build_worktable(keys_w, nq, W, nlist, qg, qg, idx->d_list_len.as<int64_t>(), idx->code_size, wt, s));
        }
        if (kind == KNHIP_IVF_PQ) {
            const int mode = !is_l2 ? PQ_LUT_IP : (idx->use_precomp ? PQ_LUT_PRECOMP : PQ_LUT_RESIDUAL);
            PqScanArgs a{};
            a.codes_skew = idx->rows2.as<uint4>();
            a.list_sblk_off = idx->d_list_blk_off2.as<int64_t>();
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
            a.coarse_dis = cdis_w;
            a.items = wt.items;
            a.pairs = wt.pairs;
            a.nitems_dev = wt.nitems;
            a.bitset = d_bitset;
            a.bitset_nbits = nbits;
            a.gthr = ws->gthr.as<float>();
            a.nslot = W;
            a.k = 1;
            a.item_lo = nullptr;
            a.item_hi = wt.nitems;
            a.dump = ws->dump.as<float>();
            a.dump_stride = ncol;
            a.dump_by_row = 1;
            HIP_TRY(launch_pq_scan_v2(a, is_l2, true, items_bound, s));
        } else {
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
            a.coarse_dis = cdis_w;
            a.items = wt.items;
            a.pairs = wt.pairs;
            a.nitems_dev = wt.nitems;
            a.bitset = d_bitset;
            a.bitset_nbits = nbits;
            a.gthr = ws->gthr.as<float>();
            a.nslot = W;
            a.k = 1;
            a.dump = ws->dump.as<float>();
            a.dump_stride = ncol;
            HIP_TRY(launch_sq_scan(a, is_l2, items_bound, s));
        }
        return KNHIP_OK;
    };
    HIP_TRY(ws->rg_cnt.reserve((size_t)nq * nprobe * sizeof(int32_t)));
    if (kind == KNHIP_IVF_PQ) {
        const int M = idx->desc.pq_m;
        const int mode = !is_l2 ? PQ_LUT_IP : (idx->use_precomp ? PQ_LUT_PRECOMP : PQ_LUT_RESIDUAL);
        if (mode != PQ_LUT_RESIDUAL) {
            HIP_TRY(ws->t2t.reserve((size_t)nq * 256 * M * sizeof(float)));
            HIP_TRY(launch_pq_query_table(d_q, idx->cb.as<float>(), d, M, nq, ws->t2t.as<float>(), s));
        }
    }
    bool counted = false;
    if (kind != KNHIP_BRUTE_FORCE) {
        const bool waves = all_lists && max_empty > 0 && nprobe > 128 && !env_search().range_no_waves;
        if (!waves) {
            if (int rc = scan_dump(ws->keys.as<int64_t>(), ws->cdis.as<float>(), nprobe)) return rc;
            if (all_lists) {
                idx->last_range_ranks = nprobe;
            }
        } else {
            // rank waves: 64 coarse ranks first, doubling; a query leaves once its run of empty lists reaches max_empty
            HIP_TRY(ws->rg_state.reserve(((size_t)nq * 2 + 1) * sizeof(int32_t)));
            HIP_TRY(hipMemsetAsync(ws->rg_state.p, 0, ((size_t)nq * 2 + 1) * sizeof(int32_t), s));
            HIP_TRY(hipMemsetAsync(ws->rg_cnt.p, 0, (size_t)nq * nprobe * sizeof(int32_t), s));
            int32_t* qstate = ws->rg_state.as<int32_t>();
            int32_t* alive = qstate + 2 * nq;
            int r0 = 0, W = std::max(64, 2 * std::min(max_empty, nprobe));
            while (r0 < nprobe) {
                const int Wc = std::min(W, nprobe - r0);
                HIP_TRY(ws->rg_keys_w.reserve((size_t)nq * Wc * sizeof(int64_t)));
                HIP_TRY(ws->rg_cdis_w.reserve((size_t)nq * Wc * sizeof(float)));
                HIP_TRY(launch_range_wave_gather(ws->keys.as<int64_t>(), ws->cdis.as<float>(), nq, nprobe, r0, Wc, qstate,
                                                 ws->rg_keys_w.as<int64_t>(), ws->rg_cdis_w.as<float>(), s));
                if (int rc = scan_dump(ws->rg_keys_w.as<int64_t>(), ws->rg_cdis_w.as<float>(), Wc)) return rc;
                HIP_TRY(launch_range_count(r, nq, is_l2, ws->rg_cnt.as<int32_t>(), s, r0, Wc, qstate));
                HIP_TRY(launch_range_wave_state(ws->rg_cnt.as<int32_t>(), nq, nprobe, r0, r0 + Wc, max_empty, qstate, alive,
                                                s));
                int32_t h_alive = 0;
                HIP_TRY(hipMemcpyAsync(&h_alive, alive, sizeof(int32_t), hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
                r0 += Wc;
                W = W < nprobe ? W * 2 : W;
                if (h_alive == 0) {
                    break;
                }
            }
            idx->last_range_ranks = r0;
            counted = true;
        }
    }
    if (dump_only_out != nullptr) { // (the caller walks the dump itself: search_batch_ties)
        *dump_only_out = r;
        return KNHIP_OK;
    }
    // count -> plan -> (host: totals, bases) -> emit
    HIP_TRY(ws->rg_off.reserve((size_t)nq * nprobe * sizeof(int64_t)));
    HIP_TRY(ws->rg_tot.reserve((size_t)nq * 2 * sizeof(int64_t)));
    if (!counted) {
        HIP_TRY(launch_range_count(r, nq, is_l2, ws->rg_cnt.as<int32_t>(), s));
    }
    HIP_TRY(launch_range_plan(ws->rg_cnt.as<int32_t>(), nq, nprobe, max_empty, ws->rg_off.as<int64_t>(),
                              ws->rg_tot.as<int64_t>(), s));
    std::vector<int64_t> tot((size_t)nq), base((size_t)nq);
    HIP_TRY(hipMemcpyAsync(tot.data(), ws->rg_tot.p, (size_t)nq * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    if (out_cnt != nullptr) { // hits per (query, coarse rank): knhip_range_search_ranked
        const size_t o = out_cnt->size();
        out_cnt->resize(o + (size_t)nq * nprobe);
        HIP_TRY(hipMemcpyAsync(out_cnt->data() + o, ws->rg_cnt.p, (size_t)nq * nprobe * sizeof(int32_t),
                               hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    int64_t run = 0;
    h_lims[0] = 0;
    for (int64_t i = 0; i < nq; i++) {
        base[i] = run;
        run += tot[i];
        h_lims[i + 1] = run;
    }
    if (run > 0) {
        HIP_TRY(hipMemcpyAsync(ws->rg_tot.as<int64_t>() + nq, base.data(), (size_t)nq * sizeof(int64_t),
                               hipMemcpyHostToDevice, s));
        HIP_TRY(ws->rg_out_i.reserve((size_t)run * sizeof(int64_t)));
        HIP_TRY(ws->rg_out_d.reserve((size_t)run * sizeof(float)));
        HIP_TRY(launch_range_emit(r, nq, is_l2, ws->rg_off.as<int64_t>(), ws->rg_tot.as<int64_t>() + nq,
                                  ws->rg_out_i.as<int64_t>(), ws->rg_out_d.as<float>(), s));
        const size_t o = out_i.size();
        out_i.resize(o + (size_t)run);
        out_d.resize(o + (size_t)run);
        HIP_TRY(hipMemcpyAsync(out_i.data() + o, ws->rg_out_i.p, (size_t)run * sizeof(int64_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(out_d.data() + o, ws->rg_out_d.p, (size_t)run * sizeof(float), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    return KNHIP_OK;
}

// segments of the distance matrix range_batch works on -> ws->rg_seg (3 x nseg: column of the first row, position of the
// first id, length): the inverted lists, or 8192-row pieces of a brute-force base
static int range_segments(const knhip_index* idx, hipStream_t s, const int64_t** d_seg_out, int64_t* nseg_out,
                          int64_t* ncol_out) {
    std::lock_guard<std::mutex> lk(idx->mu);
    if (idx->rg_seg_nseg >= 0 && idx->rg_seg_dev.p != nullptr) { // (built once per list layout)
        *d_seg_out = idx->rg_seg_dev.as<int64_t>();
        *nseg_out = idx->rg_seg_nseg;
        *ncol_out = idx->rg_seg_ncol;
        return KNHIP_OK;
    }
    const int kind = idx->desc.kind;
    std::vector<int64_t> seg;
    int64_t nseg = 0, ncol = 0;
    if (kind == KNHIP_BRUTE_FORCE) {
        const int64_t SEG = 8192;
        ncol = idx->ntotal;
        nseg = (ncol + SEG - 1) / SEG;
        seg.resize((size_t)3 * nseg);
        for (int64_t i = 0; i < nseg; i++) {
            seg[i] = seg[nseg + i] = i * SEG;
            seg[2 * nseg + i] = std::min(SEG, ncol - i * SEG);
        }
    } else {
        nseg = idx->nlist;
        seg.resize((size_t)3 * nseg);
        int64_t blk = 0;
        for (int64_t l = 0; l < nseg; l++) {
            seg[l] = kind == KNHIP_IVF_FLAT ? blk * 64 : idx->h_list_row_off[l];
            seg[nseg + l] = idx->h_list_row_off[l];
            seg[2 * nseg + l] = idx->h_list_len[l];
            blk += (idx->h_list_len[l] + 63) / 64;
        }
        ncol = kind == KNHIP_IVF_FLAT ? blk * 64 : idx->ntotal;
    }
    HIP_TRY(idx->rg_seg_dev.reserve(std::max<size_t>(seg.size(), 1) * sizeof(int64_t)));
    HIP_TRY(hipMemcpyAsync(idx->rg_seg_dev.p, seg.data(), seg.size() * sizeof(int64_t), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s)); // (the host vector is released on return)
    idx->rg_seg_nseg = nseg;
    idx->rg_seg_ncol = ncol;
    *d_seg_out = idx->rg_seg_dev.as<int64_t>();
    *nseg_out = nseg;
    *ncol_out = ncol;
    return KNHIP_OK;
}

// ---- Search() with the reference's admission rule at the k-th boundary ---------------------------------------------------
// The reference keeps its k best in a heap with STRICT-improve admission (HeapResultHandler::add_result,
// thirdparty/faiss/faiss/impl/ResultHandler.h:258-279: a candidate enters only if it beats the current k-th) and
// replaces the heap's top, which among equal distances is the one heap_replace_top's cmp2 order puts there
// (utils/Heap.h:113-151, utils/ordered_key_value.h:51, 74: the largest id for L2 / CMax, the smallest for IP / CMin);
// candidates arrive in scan order = probe rank, then storage position (IndexIVF.cpp:642-655; IndexFlat: row order).  With
// v the final k-th distance this is equivalent to (tests/test_tie_rule.py replays the heap against it):
//     a candidate tied with v is ELIGIBLE iff it is among the first k arrivals with distance <= v (L2; >= v for IP);
//     result = canonical top-k of {every candidate better than v} U {eligible ties}.
// The canonical pipeline already has v and every better candidate; it is run for k + 1 results, and only a query whose
// (k + 1)-th entry ties with its k-th (a tied candidate was left out) needs the arrival order: its probed lists are
// scanned once more in dump mode (the dump pass of range_batch over the search's own coarse assignment), one workgroup
// per query walks the dump in scan order, keeps the ties among the first k arrivals and writes the rule's answer over
// the query's row (range.hip::tie_apply_kernel) -- all on the device.  The host reads ONE 4-byte count per batch (how
// many queries are flagged: it sizes the dump) -- KNHIP_TIES=canonical skips even that and returns the canonical answer
// (the licensed deviation of include/knhip.h).
// Not covered: k = 1024 (no room for the (k + 1)-th result), brute force with k >= 100 (the reference switches to a
// reservoir, ResultHandler.h:719-728, whose boundary ties depend on its partition steps), lists sharded over several
// indexes (every shard resolves its own candidates; the merge is canonical).
static bool ties_reference_mode() { return !env_search().ties_canonical; }

extern "C" int knhip_ties_rule_applies(int32_t kind, int32_t k) {
    const bool reservoir = kind == KNHIP_BRUTE_FORCE && k >= 100;
    return (ties_reference_mode() && !reservoir && k + 1 <= KN_MAX_K) ? 1 : 0;
}

// The first k arrivals with distance <= v (>= v for IP) of every flagged query, in THIS index's scan order (probe rank, then
// storage position; brute force: row order): arr_d / arr_i [nflag][k], arr_n [nflag] (how many arrived: only min(k, .) are
// stored), arr_key [nflag][k] (optional) = each arrival's place in the scan order, comparable across the shards of a group.
// flagged[f] = row of the query in (d_q, src_keys, src_cdis, can_d [.][k + 1]); v = can_d[row][k - 1].
static int tie_arrivals(const knhip_index* idx, Workspace* ws, const float* d_q, const int32_t* flagged, int32_t nflag,
                        const float* can_d, int k, int nprobe, const int64_t* src_keys, const float* src_cdis,
                        const uint8_t* d_bitset, int64_t nbits, int64_t key_base, float* arr_d, int64_t* arr_i,
                        int64_t* arr_key, int64_t* arr_n, hipStream_t s) {
    const int kind = idx->desc.kind;
    const bool is_l2 = idx->is_l2;
    const bool trace = env_search().ties_trace;
    int64_t nseg = 0, ncol = 0;
    const int64_t* d_seg = nullptr;
    if (int rc = range_segments(idx, s, &d_seg, &nseg, &ncol)) return rc;
    const int np = kind == KNHIP_BRUTE_FORCE ? 0 : nprobe;
    if (kind != KNHIP_BRUTE_FORCE && (src_keys == nullptr || src_cdis == nullptr)) {
        return fail(KNHIP_ERR_INVALID_ARGS, "tie arrivals: the coarse assignment of the search is needed");
    }
    HIP_TRY(ws->tie_q.reserve((size_t)nflag * idx->d * sizeof(float)));
    if (src_keys != nullptr) {
        HIP_TRY(ws->tie_keys.reserve((size_t)nflag * nprobe * sizeof(int64_t)));
        HIP_TRY(ws->tie_cdis.reserve((size_t)nflag * nprobe * sizeof(float)));
    }
    HIP_TRY(ws->tie_r.reserve((size_t)nflag * sizeof(float)));
    HIP_TRY(launch_tie_gather(flagged, nflag, d_q, idx->d, src_keys, src_cdis, nprobe, ws->tie_q.as<float>(),
                              ws->tie_keys.as<int64_t>(), ws->tie_cdis.as<float>(), can_d, k, ws->tie_r.as<float>(), s));
    // queries per round: the dump matrix [round][ncol] stays below 2 GiB
    int64_t qb = std::max<int64_t>(1, (int64_t)((2ull << 30) / ((size_t)std::max<int64_t>(ncol, 1) * 4)));
    qb = std::max<int64_t>(1, std::min<int64_t>(qb, (int64_t)0x7fffffff / std::max<int64_t>(nseg, 1)));
    for (int64_t f0 = 0; f0 < nflag; f0 += qb) {
        const int64_t n = std::min<int64_t>(qb, nflag - f0);
        if (trace) fprintf(stderr, "[ties] round f0=%lld n=%lld ncol=%lld nseg=%lld np=%d\n", (long long)f0, (long long)n, (long long)ncol, (long long)nseg, np);
        std::vector<int64_t> lims_unused((size_t)n + 1), hi_unused;
        std::vector<float> hd_unused;
        RangeArgs r{};
        if (int rc = range_batch(idx, ws, ws->tie_q.as<float>() + f0 * idx->d, n, 0.f, 0, d_bitset, nbits, d_seg, nseg, ncol,
                                 lims_unused.data(), hi_unused, hd_unused, s, nullptr, np,
                                 src_keys ? ws->tie_keys.as<int64_t>() + f0 * nprobe : nullptr,
                                 src_keys ? ws->tie_cdis.as<float>() + f0 * nprobe : nullptr, &r)) {
            return rc;
        }
        // count per (query, rank) -> offsets -> capped emit (all parallel over the ranks)
        r.radius_q = ws->tie_r.as<float>() + f0;
        r.inclusive = 1;
        HIP_TRY(ws->rg_cnt.reserve((size_t)n * r.nprobe * sizeof(int32_t)));
        HIP_TRY(ws->rg_off.reserve((size_t)n * r.nprobe * sizeof(int64_t)));
        HIP_TRY(ws->rg_tot.reserve((size_t)n * 2 * sizeof(int64_t)));
        HIP_TRY(launch_range_count(r, n, is_l2, ws->rg_cnt.as<int32_t>(), s));
        HIP_TRY(launch_range_plan(ws->rg_cnt.as<int32_t>(), n, r.nprobe, 0, ws->rg_off.as<int64_t>(), ws->rg_tot.as<int64_t>(), s));
        HIP_TRY(launch_range_emit(r, n, is_l2, ws->rg_off.as<int64_t>(), nullptr, arr_i + f0 * k, arr_d + f0 * k, s, k,
                                  arr_key != nullptr ? arr_key + f0 * k : nullptr, key_base));
        HIP_TRY(hipMemcpyAsync(arr_n + f0, ws->rg_tot.as<int64_t>(), (size_t)n * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
    }
    return KNHIP_OK;
}

__global__ void publish_word_kernel(const int32_t* __restrict__ src, volatile int32_t* host_word, int32_t seq) {
    host_word[0] = *src;
    __threadfence_system();
    host_word[1] = seq;
}

// *out = the device word `d_word` as of this point of the stream, without draining the host's view of the stream through
// hipStreamSynchronize (see Workspace::h_word).  The stream is queried now and then while spinning: a launch that failed
// would otherwise never publish
static int read_word_now(Workspace* ws, const int32_t* d_word, int32_t* out, hipStream_t s) {
    if (ws->h_word == nullptr) {
        void* p = nullptr;
        HIP_TRY(hipHostMalloc(&p, 16 * sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(p, 0, 16 * sizeof(int32_t));
        ws->h_word = static_cast<volatile int32_t*>(p);
    }
    const int32_t seq = ++ws->word_seq;
    hipLaunchKernelGGL(publish_word_kernel, dim3(1), dim3(1), 0, s, d_word, ws->h_word, seq);
    HIP_TRY(hipGetLastError());
    for (uint64_t spins = 0;; spins++) {
        if (ws->h_word[1] == seq) {
            break;
        }
        if ((spins & 0xfff) == 0xfff) {
            const hipError_t q = hipStreamQuery(s);
            if (q == hipSuccess) { // (the stream has drained: the word is there, or never will be)
                if (ws->h_word[1] != seq) {
                    return fail(KNHIP_ERR_HIP_RUNTIME, "a count published by the device did not arrive");
                }
                break;
            }
            if (q != hipErrorNotReady) {
                HIP_TRY(q);
            }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    *out = ws->h_word[0];
    return KNHIP_OK;
}

} // extern "C"

namespace knhip_host {
int search_batch_ties(const knhip_index* idx, Workspace* ws, const float* d_q, int64_t nq, int k, int nprobe,
                             const uint8_t* d_bitset, int64_t nbits, int64_t* d_out_i, float* d_out_d, hipStream_t s,
                             const int64_t* pre_keys, const float* pre_cdis) {
    const int kind = idx->desc.kind;
    if (!knhip_ties_rule_applies(kind, k)) {
        return search_batch(idx, ws, d_q, nq, k, nprobe, d_bitset, nbits, d_out_i, d_out_d, s, pre_keys, pre_cdis);
    }
    const bool trace = env_search().ties_trace;
    const int kk = k + 1;
    if (trace) fprintf(stderr, "[ties] search nq=%lld k=%d nprobe=%d kind=%d\n", (long long)nq, k, nprobe, kind);
    const bool is_l2 = idx->is_l2;
    HIP_TRY(ws->tie_d.reserve((size_t)nq * kk * sizeof(float)));
    HIP_TRY(ws->tie_i.reserve((size_t)nq * kk * sizeof(int64_t)));
    HIP_TRY(ws->tie_flag.reserve(((size_t)nq + 1) * sizeof(int32_t)));
    if (int rc = search_batch(idx, ws, d_q, nq, kk, nprobe, d_bitset, nbits, ws->tie_i.as<int64_t>(), ws->tie_d.as<float>(), s,
                              pre_keys, pre_cdis)) {
        return rc;
    }
    if (trace) fprintf(stderr, "[ties] searched\n");
    StageTimer t_ties(idx, s, KNHIP_STAGE_TIES); // (detection, read-back and -- for the flagged queries -- dump pass + rule)
    int32_t* flagged = ws->tie_flag.as<int32_t>();
    int32_t* nflag_dev = flagged + nq;
    HIP_TRY(hipMemsetAsync(nflag_dev, 0, sizeof(int32_t), s));
    HIP_TRY(launch_tie_detect(ws->tie_d.as<float>(), ws->tie_i.as<int64_t>(), nq, k, d_out_d, d_out_i, flagged, nflag_dev, s));
    int32_t nflag = 0;
    if (int rc = read_word_now(ws, nflag_dev, &nflag, s)) return rc;
    if (trace) fprintf(stderr, "[ties] flagged %d\n", nflag);
    if (nflag <= 0) {
        return KNHIP_OK;
    }
    // ---- flagged queries, on the device: every distance of their probed lists dumped by the range-search pass, the first
    // k arrivals with distance <= v taken in scan order (tie_arrivals), then one workgroup per query writes the reference's
    // answer over the query's row of the output.  Nothing comes back to the host.
    const int64_t* src_keys = pre_keys != nullptr ? pre_keys : (kind != KNHIP_BRUTE_FORCE ? ws->keys.as<int64_t>() : nullptr);
    const float* src_cdis = pre_cdis != nullptr ? pre_cdis : (kind != KNHIP_BRUTE_FORCE ? ws->cdis.as<float>() : nullptr);
    HIP_TRY(ws->tie_arr_d.reserve((size_t)nflag * k * sizeof(float)));
    HIP_TRY(ws->tie_arr_i.reserve((size_t)nflag * k * sizeof(int64_t)));
    HIP_TRY(ws->tie_arr_n.reserve((size_t)nflag * sizeof(int64_t)));
    if (int rc = tie_arrivals(idx, ws, d_q, flagged, nflag, ws->tie_d.as<float>(), k, nprobe, src_keys, src_cdis, d_bitset,
                              nbits, 0, ws->tie_arr_d.as<float>(), ws->tie_arr_i.as<int64_t>(), nullptr,
                              ws->tie_arr_n.as<int64_t>(), s)) {
        return rc;
    }
    HIP_TRY(launch_tie_resolve(flagged, nflag, 1, ws->tie_d.as<float>(), ws->tie_i.as<int64_t>(), k, is_l2,
                               ws->tie_arr_d.as<float>(), ws->tie_arr_i.as<int64_t>(), nullptr, ws->tie_arr_n.as<int64_t>(),
                               nflag, d_out_d, d_out_i,
                               reinterpret_cast<int32_t*>(idx->coarse_fail_dev.as<unsigned long long>() + 5), s));
    if (trace) fprintf(stderr, "[ties] applied\n");
    {
        std::lock_guard<std::mutex> lk(idx->mu);
        idx->tie_queries += nflag;
    }
    return KNHIP_OK;
}
} // namespace knhip_host

extern "C" {

// ---- the same rule for a list-sharded index: the pieces a shard host strings together (include/knhip.h) ---------------------
int knhip_search_canonical_device(const knhip_index* idx, const float* d_queries, int64_t nq, int32_t k, int32_t nprobe,
                                  const int64_t* d_keys, const float* d_coarse_dis, const uint8_t* d_bitset,
                                  int64_t bitset_nbits, int64_t* d_out_ids, float* d_out_dist, void* stream) {
    if (int rc = check_index(idx)) return rc;
    const int32_t nprobe_in = nprobe;
    if (int rc = validate_search(idx, nq, k, nprobe)) return rc;
    const bool pre = d_keys != nullptr || d_coarse_dis != nullptr;
    if (pre && (idx->desc.kind == KNHIP_BRUTE_FORCE || !d_keys || !d_coarse_dis || nprobe != nprobe_in)) {
        return fail(KNHIP_ERR_INVALID_ARGS, "search_canonical: a coarse assignment needs an IVF index, both arrays and nprobe <= nlist");
    }
    if (nq == 0) {
        return KNHIP_OK;
    }
    if (!d_queries || !d_out_ids || !d_out_dist) {
        return fail(KNHIP_ERR_INVALID_ARGS, "null query/output pointer");
    }
    DeviceGuard g(idx->desc.device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    Workspace* ws = acquire_ws(idx, stream, true);
    std::lock_guard<std::mutex> ws_lock(ws->mu);
    const int64_t qb = query_batch(idx, nq, k, nprobe);
    for (int64_t q0 = 0; q0 < nq; q0 += qb) {
        const int64_t n = std::min(qb, nq - q0);
        if (int rc = search_batch(idx, ws, d_queries + q0 * idx->d, n, k, nprobe, d_bitset, bitset_nbits, d_out_ids + q0 * k,
                                  d_out_dist + q0 * k, s, pre ? d_keys + q0 * nprobe : nullptr,
                                  pre ? d_coarse_dis + q0 * nprobe : nullptr)) {
            return rc;
        }
    }
    return KNHIP_OK;
}

int knhip_tie_flag_device(const float* d_can_dist, const int64_t* d_can_ids, int64_t nq, int32_t k, float* d_out_dist,
                          int64_t* d_out_ids, int32_t* d_flagged, int32_t* nflag_out, void* stream) {
    if (nq < 0 || k <= 0 || k + 1 > KN_MAX_K || !d_can_dist || !d_can_ids || !d_out_dist || !d_out_ids || !d_flagged ||
        !nflag_out) {
        return fail(KNHIP_ERR_INVALID_ARGS, "tie_flag: bad arguments");
    }
    *nflag_out = 0;
    if (nq == 0) {
        return KNHIP_OK;
    }
    if (nq > 0x3fffffff) {
        return fail(KNHIP_ERR_INVALID_ARGS, "tie_flag: too many queries");
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    // d_flagged [2 nq + 1]: the sorted list, the raw (atomic-order) list behind it, the count
    int32_t* raw = d_flagged + nq;
    int32_t* cnt = d_flagged + 2 * nq;
    HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(int32_t), s));
    HIP_TRY(launch_tie_detect(d_can_dist, d_can_ids, nq, k, d_out_dist, d_out_ids, raw, cnt, s));
    int32_t nflag = 0;
    HIP_TRY(hipMemcpyAsync(&nflag, cnt, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(launch_tie_sort_flags(raw, nflag, d_flagged, s));
    *nflag_out = nflag;
    return KNHIP_OK;
}

int knhip_tie_arrivals_device(const knhip_index* idx, const float* d_queries, const int32_t* d_flagged, int32_t nflag,
                              const float* d_can_dist, int32_t k, int32_t nprobe, const int64_t* d_keys,
                              const float* d_coarse_dis, const uint8_t* d_bitset, int64_t bitset_nbits, int64_t key_base,
                              float* d_arr_dist, int64_t* d_arr_ids, int64_t* d_arr_key, int64_t* d_arr_n, void* stream) {
    if (int rc = check_index(idx)) return rc;
    if (nflag < 0 || k <= 0 || k + 1 > KN_MAX_K || !d_queries || !d_flagged || !d_can_dist || !d_arr_dist || !d_arr_ids ||
        !d_arr_key || !d_arr_n) {
        return fail(KNHIP_ERR_INVALID_ARGS, "tie_arrivals: bad arguments");
    }
    if (nflag == 0) {
        return KNHIP_OK;
    }
    DeviceGuard g(idx->desc.device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!idx->has_data || idx->ntotal == 0) { // (a shard without rows: nothing arrives)
        HIP_TRY(hipMemsetAsync(d_arr_n, 0, (size_t)nflag * sizeof(int64_t), s));
        return KNHIP_OK;
    }
    int32_t np = nprobe;
    if (idx->desc.kind != KNHIP_BRUTE_FORCE) {
        if (np <= 0 || np > idx->nlist) {
            return fail(KNHIP_ERR_INVALID_ARGS, "tie_arrivals: nprobe out of range");
        }
    }
    Workspace* ws = acquire_ws(idx, stream, true);
    std::lock_guard<std::mutex> ws_lock(ws->mu);
    return tie_arrivals(idx, ws, d_queries, d_flagged, nflag, d_can_dist, k, np, d_keys, d_coarse_dis, d_bitset, bitset_nbits,
                        key_base, d_arr_dist, d_arr_ids, d_arr_key, d_arr_n, s);
}

int knhip_tie_resolve_device(int32_t metric, int32_t nshards, const int32_t* d_flagged, int32_t nflag, int32_t k,
                             const float* d_can_dist, const int64_t* d_can_ids, const float* d_arr_dist,
                             const int64_t* d_arr_ids, const int64_t* d_arr_key, const int64_t* d_arr_n, float* d_out_dist,
                             int64_t* d_out_ids, void* stream) {
    if (nshards <= 0 || nflag < 0 || k <= 0 || k + 1 > KN_MAX_K || (metric != KNHIP_L2 && metric != KNHIP_IP) ||
        !d_flagged || !d_can_dist || !d_can_ids || !d_arr_dist || !d_arr_ids || !d_arr_key || !d_arr_n || !d_out_dist ||
        !d_out_ids) {
        return fail(KNHIP_ERR_INVALID_ARGS, "tie_resolve: bad arguments");
    }
    HIP_TRY(launch_tie_resolve(d_flagged, nflag, nshards, d_can_dist, d_can_ids, k, metric == KNHIP_L2, d_arr_dist, d_arr_ids,
                               d_arr_key, d_arr_n, nflag, d_out_dist, d_out_ids, nullptr, static_cast<hipStream_t>(stream)));
    return KNHIP_OK;
}

// host form of flag + resolve (results merged on the CPU: knhip_merge_topk_host's companion).  arr_* [nshards][nq][k] /
// arr_n [nshards][nq] indexed by the QUERY (rows of unflagged queries are not read); flagged_out (nullable) [nq] 0 / 1.
int knhip_tie_flag_host(const float* can_dist, const int64_t* can_ids, int64_t nq, int32_t k, float* out_dist,
                        int64_t* out_ids, uint8_t* flagged_out) {
    if (nq < 0 || k <= 0 || !can_dist || !can_ids || !out_dist || !out_ids) {
        return fail(KNHIP_ERR_INVALID_ARGS, "tie_flag_host: bad arguments");
    }
    const int kk = k + 1;
    for (int64_t q = 0; q < nq; q++) {
        for (int j = 0; j < k; j++) {
            out_dist[q * k + j] = can_dist[q * kk + j];
            out_ids[q * k + j] = can_ids[q * kk + j];
        }
        uint32_t a, b;
        std::memcpy(&a, &can_dist[q * kk + k], 4);
        std::memcpy(&b, &can_dist[q * kk + k - 1], 4);
        if (flagged_out) {
            flagged_out[q] = (can_ids[q * kk + k] >= 0 && can_ids[q * kk + k - 1] >= 0 && a == b) ? 1 : 0;
        }
    }
    return KNHIP_OK;
}

int knhip_tie_resolve_host(int32_t metric, int32_t nshards, int64_t nq, int32_t k, const uint8_t* flagged,
                           const float* can_dist, const int64_t* can_ids, const float* arr_dist, const int64_t* arr_ids,
                           const int64_t* arr_key, const int64_t* arr_n, float* out_dist, int64_t* out_ids) {
    if (nshards <= 0 || nq < 0 || k <= 0 || (metric != KNHIP_L2 && metric != KNHIP_IP) || !flagged || !can_dist || !can_ids ||
        !arr_dist || !arr_ids || !arr_key || !arr_n || !out_dist || !out_ids) {
        return fail(KNHIP_ERR_INVALID_ARGS, "tie_resolve_host: bad arguments");
    }
    const bool l2 = metric == KNHIP_L2;
    const int kk = k + 1;
    struct Arr {
        int64_t key;
        float d;
        int64_t id;
    };
    std::vector<Arr> arr;
    std::vector<std::pair<float, int64_t>> pool;
    for (int64_t q = 0; q < nq; q++) {
        if (!flagged[q]) {
            continue;
        }
        const float v = can_dist[q * kk + k - 1];
        pool.clear();
        size_t nvalid = 0;
        for (int j = 0; j < k; j++) {
            if (can_ids[q * kk + j] >= 0) {
                nvalid++;
                if (can_dist[q * kk + j] != v) {
                    pool.emplace_back(can_dist[q * kk + j], can_ids[q * kk + j]);
                }
            }
        }
        arr.clear();
        for (int sh = 0; sh < nshards; sh++) {
            const int64_t cnt = std::min<int64_t>(k, arr_n[(int64_t)sh * nq + q]);
            const int64_t at = ((int64_t)sh * nq + q) * k;
            for (int64_t e = 0; e < cnt; e++) {
                arr.push_back({arr_key[at + e], arr_dist[at + e], arr_ids[at + e]});
            }
        }
        std::stable_sort(arr.begin(), arr.end(), [](const Arr& a, const Arr& b) { return a.key < b.key; });
        for (size_t e = 0; e < arr.size() && e < (size_t)k; e++) { // the first k arrivals overall
            if (arr[e].d == v) {
                pool.emplace_back(v, arr[e].id);
            }
        }
        if (pool.size() < nvalid) {
            continue; // (as the device kernel: the canonical row stays)
        }
        std::sort(pool.begin(), pool.end(), [l2](const std::pair<float, int64_t>& a, const std::pair<float, int64_t>& b) {
            if (l2) {
                return a.first < b.first || (a.first == b.first && a.second < b.second);
            }
            return a.first > b.first || (a.first == b.first && a.second > b.second);
        });
        for (int j = 0; j < k && (size_t)j < pool.size(); j++) {
            out_dist[q * k + j] = pool[j].first;
            out_ids[q * k + j] = pool[j].second;
        }
    }
    return KNHIP_OK;
}

static int range_search_impl(const knhip_index* idx, const float* queries, int64_t nq, float radius,
                             int32_t max_empty_result_buckets, const uint8_t* bitset, int64_t bitset_nbits, int64_t* lims,
                             int64_t** out_ids, float** out_dist, int32_t** out_rank_counts) {
    if (int rc = check_index(idx)) return rc;
    if (out_rank_counts) {
        *out_rank_counts = nullptr;
    }
    if (nq < 0 || max_empty_result_buckets < 0 || !lims || !out_ids || !out_dist) {
        return fail(KNHIP_ERR_INVALID_ARGS, "bad range search arguments");
    }
    *out_ids = nullptr;
    *out_dist = nullptr;
    lims[0] = 0;
    if (nq == 0) {
        return KNHIP_OK;
    }
    if (!queries) {
        return fail(KNHIP_ERR_INVALID_ARGS, "null query pointer");
    }
    if (!idx->has_data) {
        return fail(KNHIP_ERR_EMPTY_INDEX, "index holds no vectors");
    }
    const int kind = idx->desc.kind;
    // (IVF_PQ: m = 32 dumps through the stream16 scan, other code widths through the plain ADC dump kernel of range.hip)
    if (kind != KNHIP_BRUTE_FORCE && (size_t)idx->nlist > row_select_max_k()) {
        return fail(KNHIP_ERR_NOT_IMPLEMENTED, "range search probes every list: nlist > 65536 is not supported");
    }
    DeviceGuard g(idx->desc.device);
    hipStream_t s = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    Workspace* ws = acquire_ws(idx, nullptr, false);
    std::vector<int64_t> res_i;
    std::vector<float> res_d;
    std::vector<int32_t> res_cnt;
    auto run = [&]() -> int {
        int64_t nseg = 0, ncol = 0;
        const int64_t* d_seg = nullptr;
        if (int r = range_segments(idx, s, &d_seg, &nseg, &ncol)) return r;
        const size_t qbytes = (size_t)nq * idx->d * sizeof(float);
        HIP_TRY(ws->h_queries.reserve(qbytes));
        HIP_TRY(hipMemcpyAsync(ws->h_queries.p, queries, qbytes, hipMemcpyHostToDevice, s));
        const uint8_t* d_bitset = nullptr;
        if (bitset && bitset_nbits > 0) {
            const size_t bb = (size_t)((bitset_nbits + 7) / 8);
            HIP_TRY(ws->h_bitset.reserve(bb));
            HIP_TRY(hipMemcpyAsync(ws->h_bitset.p, bitset, bb, hipMemcpyHostToDevice, s));
            d_bitset = ws->h_bitset.as<uint8_t>();
        }
        // queries per batch: the distance matrix stays below 2 GiB
        // (and one workgroup per (query, segment) must fit a launch)
        int64_t qb = std::max<int64_t>(1, std::min<int64_t>(nq, (int64_t)((2ull << 30) / ((size_t)std::max<int64_t>(ncol, 1) * 4))));
        qb = std::max<int64_t>(1, std::min<int64_t>(qb, (int64_t)0x7fffffff / std::max<int64_t>(nseg, 1)));
        std::vector<int64_t> rel((size_t)qb + 1);
        for (int64_t q0 = 0; q0 < nq; q0 += qb) {
            const int64_t n = std::min(qb, nq - q0);
            if (int r = range_batch(idx, ws, ws->h_queries.as<float>() + q0 * idx->d, n, radius, max_empty_result_buckets,
                                    d_bitset, bitset_nbits, d_seg, nseg, ncol, rel.data(), res_i,
                                    res_d, s, nullptr, 0, nullptr, nullptr, nullptr, out_rank_counts ? &res_cnt : nullptr)) {
                return r;
            }
            for (int64_t i = 0; i < n; i++) {
                lims[q0 + i + 1] = lims[q0] + rel[i + 1];
            }
        }
        return KNHIP_OK;
    };
    int rc = run();
    if (rc != KNHIP_OK) {
        (void)hipStreamSynchronize(s);
    }
    release_ws(idx, ws);
    (void)hipStreamDestroy(s);
    if (rc != KNHIP_OK) {
        return rc;
    }
    const size_t n = res_i.size();
    *out_ids = static_cast<int64_t*>(std::malloc(sizeof(int64_t) * (n + 1)));
    *out_dist = static_cast<float*>(std::malloc(sizeof(float) * (n + 1)));
    if (!*out_ids || !*out_dist) {
        std::free(*out_ids);
        std::free(*out_dist);
        *out_ids = nullptr;
        *out_dist = nullptr;
        return fail(KNHIP_ERR_OUT_OF_MEMORY, "host allocation of the range result failed");
    }
    std::memcpy(*out_ids, res_i.data(), sizeof(int64_t) * n);
    std::memcpy(*out_dist, res_d.data(), sizeof(float) * n);
    if (out_rank_counts) {
        *out_rank_counts = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * (res_cnt.size() + 1)));
        if (!*out_rank_counts) {
            std::free(*out_ids);
            std::free(*out_dist);
            *out_ids = nullptr;
            *out_dist = nullptr;
            return fail(KNHIP_ERR_OUT_OF_MEMORY, "host allocation of the range result failed");
        }
        std::memcpy(*out_rank_counts, res_cnt.data(), sizeof(int32_t) * res_cnt.size());
    }
    return KNHIP_OK;
}

int knhip_range_search(const knhip_index* idx, const float* queries, int64_t nq, float radius,
                       int32_t max_empty_result_buckets, const uint8_t* bitset, int64_t bitset_nbits, int64_t* lims,
                       int64_t** out_ids, float** out_dist) {
    return range_search_impl(idx, queries, nq, radius, max_empty_result_buckets, bitset, bitset_nbits, lims, out_ids, out_dist,
                             nullptr);
}

int knhip_range_search_ranked(const knhip_index* idx, const float* queries, int64_t nq, float radius, const uint8_t* bitset,
                              int64_t bitset_nbits, int64_t* lims, int64_t** out_ids, float** out_dist,
                              int32_t** out_rank_counts) {
    if (!out_rank_counts) {
        return fail(KNHIP_ERR_INVALID_ARGS, "range_search_ranked: null count pointer");
    }
    if (int rc = check_index(idx)) return rc;
    if (idx->desc.kind == KNHIP_BRUTE_FORCE) {
        return fail(KNHIP_ERR_INVALID_ARGS, "range_search_ranked: an IVF index (a brute-force base has no coarse ranks)");
    }
    return range_search_impl(idx, queries, nq, radius, 0, bitset, bitset_nbits, lims, out_ids, out_dist, out_rank_counts);
}

} // extern "C"
