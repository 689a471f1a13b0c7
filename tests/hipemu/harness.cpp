// tests/hipemu/harness.cpp -- drives the emulated kernels of pq_filter.hip (+ the sample plan, tau and finish kernels of
// mfma_scan.hip) through the IVF-PQ prefilter pipeline on host memory; test infrastructure only (see hip/hip_runtime.h).
//
// What runs as emulated kernel SOURCE: pq_stream16r, pq_psum, pqf_query_table, pqf_prepare, pqf_kernel (sample and
// filter mode), ms_sample_plan, ms_tau, mscan_finish (KIND 2).  What the harness does on the host instead of the
// product's kernels (worktable.hip / topk.hip are not part of the emulation): grouping the (query, probe) pairs into
// units of <= 8 per list, and the k best of each query's sample row.
#include "common.h"
#include "kernels.h"

#include <algorithm>
#include <cfloat>
#include <vector>

using namespace knhip;

namespace {

// pairs of the given class grouped by list -> units of at most 8 pairs (ascending list order, as the work table)
void make_units(const int64_t* keys, const int32_t* sample_off, int64_t nq, int nprobe, int64_t nlist, const int64_t* list_len,
                bool sampled_only, std::vector<KnPair>& pairs, std::vector<KnItem>& units) {
    std::vector<std::vector<KnPair>> by_list((size_t)nlist);
    for (int64_t q = 0; q < nq; q++) {
        for (int s = 0; s < nprobe; s++) {
            const int64_t l = keys[q * nprobe + s];
            if (l < 0 || l >= nlist || list_len[l] == 0) {
                continue;
            }
            if (sampled_only && sample_off[q * nprobe + s] < 0) {
                continue;
            }
            by_list[(size_t)l].push_back(KnPair{(int32_t)q, (int32_t)s});
        }
    }
    pairs.clear();
    units.clear();
    for (int64_t l = 0; l < nlist; l++) {
        const auto& v = by_list[(size_t)l];
        for (size_t i = 0; i < v.size(); i += 8) {
            KnItem it;
            it.list = (int32_t)l;
            it.npair = (int32_t)std::min<size_t>(8, v.size() - i);
            it.pair0 = (int64_t)pairs.size();
            units.push_back(it);
            for (size_t j = 0; j < (size_t)it.npair; j++) {
                pairs.push_back(v[i + j]);
            }
        }
    }
}

} // namespace

extern "C" int emu_pqf_search(int64_t nlist, const int64_t* list_len, const int64_t* list_row_off, const uint8_t* codes,
                              const int64_t* ids, const float* precomp /* [nlist][32][256] or null */,
                              const float* cb /* [32][256][4] */, const float* centroids, const float* xq, int64_t nq,
                              int nprobe, const int64_t* keys, const float* cdis, int k, int is_l2, int cap,
                              const uint8_t* bitset, int64_t nbits, int use_hist, int do_retry, float* out_d, int64_t* out_i,
                              int32_t* cand_cnt_out, int32_t* overflow_out, float* tau_out, int64_t* nunits_out, int32_t* poor_out) {
    const int d = 128, M = 32;
    const bool l2 = is_l2 != 0;
    // ---- index side: c-major codebook, transposed term-2 table, rotated stream, per-vector sums ----------------------
    std::vector<float4> cb_t((size_t)256 * M);
    for (int m = 0; m < M; m++) {
        for (int c = 0; c < 256; c++) {
            const float* y = cb + ((size_t)m * 256 + c) * 4;
            cb_t[(size_t)c * M + m] = make_float4(y[0], y[1], y[2], y[3]);
        }
    }
    std::vector<float> precomp_t;
    if (l2) {
        precomp_t.resize((size_t)nlist * 256 * M);
        for (int64_t l = 0; l < nlist; l++) {
            for (int m = 0; m < M; m++) {
                for (int c = 0; c < 256; c++) {
                    precomp_t[((size_t)l * 256 + c) * M + m] = precomp[((size_t)l * M + m) * 256 + c];
                }
            }
        }
    }
    std::vector<int64_t> sblk((size_t)nlist + 1, 0);
    for (int64_t l = 0; l < nlist; l++) {
        sblk[(size_t)l + 1] = sblk[(size_t)l] + pq_stream16r_blocks(list_len[l]);
    }
    std::vector<uint4> rows_r((size_t)std::max<int64_t>(sblk[(size_t)nlist], 1) * 64);
    if (launch_pq_stream16r(codes, list_row_off, list_len, sblk.data(), nlist, rows_r.data(), nullptr) != hipSuccess) return 1;
    const size_t npos = (size_t)std::max<int64_t>(sblk[(size_t)nlist], 1) * 16;
    std::vector<float> psum(npos + 4, 0.f);
    float pabs_max = 0.f;
    if (l2) {
        uint32_t* bits = reinterpret_cast<uint32_t*>(psum.data() + npos);
        if (launch_pq_psum(codes, list_row_off, list_len, sblk.data(), nlist, precomp_t.data(), psum.data(), bits, nullptr) != hipSuccess) return 2;
        std::memcpy(&pabs_max, bits, 4);
    }
    // ---- per query: half tables ----------------------------------------------------------------------------------
    std::vector<uint16_t> qh((size_t)nq * 256 * 32);
    std::vector<float> qs((size_t)nq * 4);
    if (launch_pqf_query_table(xq, reinterpret_cast<const float4*>(cb), d, nq, l2, pabs_max, qh.data(), qs.data(), nullptr) != hipSuccess) return 3;
    // ---- sample plan (emulated kernel), units of the sampled pairs -----------------------------------------------------
    const int sample = mscan_sample_rows();
    std::vector<int32_t> sample_off((size_t)nq * nprobe), n_row((size_t)nq);
    if (launch_ms_sample_plan(keys, nq, nprobe, nlist, list_len, std::max(1024, 8 * k), sample, sample_off.data(), n_row.data(), nullptr) != hipSuccess) return 4;
    std::vector<KnPair> pairs;
    std::vector<KnItem> units;
    make_units(keys, sample_off.data(), nq, nprobe, nlist, list_len, true, pairs, units);
    int64_t nunits = (int64_t)units.size();
    std::vector<P8Rec> recs((size_t)std::max<int64_t>((int64_t)nq * nprobe, 8));
    std::vector<int32_t> ctr(8 * 16, 0);
    std::vector<int32_t> cand_cnt((size_t)2 * nq + 1, 0);
    std::vector<int64_t> cand((size_t)nq * cap);
    std::vector<float> dump((size_t)nq * sample, l2 ? FLT_MAX : -FLT_MAX);
    std::vector<float> gthr((size_t)nq);
    std::vector<uint32_t> ghist((size_t)nq * 64, 0);
    std::vector<uint2> gmeta((size_t)nq);

    MScanArgs m{};
    m.list_len = list_len;
    m.list_row_off = list_row_off;
    m.ids = ids;
    m.centroids = centroids;
    m.d = d;
    m.nchunk = d / 4;
    m.queries = xq;
    m.coarse_dis = cdis;
    m.nq = nq;
    m.nslot = nprobe;
    m.units = units.data();
    m.pairs = pairs.data();
    m.nunits_dev = &nunits;
    m.gthr = gthr.data();
    m.gthr_rw = gthr.data();
    m.bitset = bitset;
    m.bitset_nbits = nbits;
    m.cand_cnt = cand_cnt.data();
    m.cand = cand.data();
    m.cap = cap;
    m.overflow = cand_cnt.data() + nq;
    m.k = k;
    m.pq_codes_r = rows_r.data();
    m.pq_sblk_off_r = sblk.data();
    m.pq_psum = psum.data();
    m.pq_qh = qh.data();
    m.pq_qs = qs.data();
    m.pq_cb_t = cb_t.data();
    m.pq_precomp_t = l2 ? precomp_t.data() : nullptr;
    m.pq_codes = codes;
    m.pq_lut_mode = l2 ? PQ_LUT_PRECOMP : PQ_LUT_IP;
    m.pq_recs = recs.data();
    m.pq_ctr = ctr.data();
    // ---- phase 1: sample pass -> tau ---------------------------------------------------------------------------------------
    {
        MScanArgs ds = m;
        ds.dump = dump.data();
        ds.dump_stride = sample;
        ds.sample_off = sample_off.data();
        if (launch_pqf(ds, l2, std::max<int64_t>(nunits, 1), nullptr) != hipSuccess) return 5;
    }
    std::vector<float> sel_d((size_t)nq * k);
    for (int64_t q = 0; q < nq; q++) {
        std::vector<float> row(dump.begin() + q * sample, dump.begin() + q * sample + n_row[(size_t)q]);
        if (l2) {
            std::sort(row.begin(), row.end());
        } else {
            std::sort(row.begin(), row.end(), [](float a, float b) { return a > b; });
        }
        for (int e = 0; e < k; e++) {
            sel_d[(size_t)q * k + e] = e < (int)row.size() ? row[(size_t)e] : (l2 ? FLT_MAX : -FLT_MAX);
        }
    }
    if (launch_ms_tau(sel_d.data(), nq, k, l2, gthr.data(), gmeta.data(), nullptr) != hipSuccess) return 6;
    for (int64_t q = 0; q < nq; q++) {
        tau_out[q] = gthr[(size_t)q];
    }
    {   // the selectivity guard's prediction (the product abandons the prefilter for a batch on it; here it is reported)
        int32_t poor[2] = {0, 0};
        if (launch_pqf_predict(dump.data(), sample, n_row.data(), gthr.data(), qs.data(), nullptr, keys, nprobe, nlist, list_len,
                               nq, cap, k, l2, poor, nullptr) != hipSuccess) return 11;
        *poor_out = poor[0];
    }
    if (use_hist) {
        m.ghist = ghist.data();
        m.gmeta = gmeta.data();
    }
    // ---- phase 2: filter over every pair ---------------------------------------------------------------------------------------
    make_units(keys, sample_off.data(), nq, nprobe, nlist, list_len, false, pairs, units);
    nunits = (int64_t)units.size();
    *nunits_out = nunits;
    m.units = units.data();
    m.pairs = pairs.data();
    if (launch_pqf(m, l2, std::max<int64_t>(nunits, 1), nullptr) != hipSuccess) return 7;
    // ---- phase 3: exact finish of the candidates ---------------------------------------------------------------------------
    unsigned long long counters[4] = {0, 0, 0, 0};
    if (launch_mscan_finish(m, 2, l2, keys, cdis, nprobe, k, out_d, out_i, counters, 1, nullptr) != hipSuccess) return 8;
    // ---- phase 4: the retry round of the queries whose list overflowed (flag 2 after pass 1): one-pair units of all
    // their probes, no histogram (the rows were counted once already), second pass of the finish kernel ---------------
    if (do_retry) {
        const int32_t* flag = cand_cnt.data() + nq;
        pairs.clear();
        units.clear();
        for (int64_t q = 0; q < nq; q++) {
            if (flag[q] != 2) {
                continue;
            }
            for (int s = 0; s < nprobe; s++) {
                const int64_t l = keys[q * nprobe + s];
                if (l < 0 || l >= nlist || list_len[l] == 0) {
                    continue;
                }
                KnItem it;
                it.list = (int32_t)l;
                it.npair = 1;
                it.pair0 = (int64_t)pairs.size();
                units.push_back(it);
                pairs.push_back(KnPair{(int32_t)q, (int32_t)s});
            }
        }
        nunits = (int64_t)units.size();
        MScanArgs r = m;
        r.units = units.data();
        r.pairs = pairs.data();
        r.unit_loop = 1;
        r.ghist = nullptr;
        r.gmeta = nullptr;
        if (nunits > 0 && launch_pqf(r, l2, nunits, nullptr) != hipSuccess) return 9;
        if (launch_mscan_finish(m, 2, l2, keys, cdis, nprobe, k, out_d, out_i, counters, 2, nullptr) != hipSuccess) return 10;
    }
    for (int64_t q = 0; q < nq; q++) {
        cand_cnt_out[q] = cand_cnt[(size_t)q];
        overflow_out[q] = cand_cnt[(size_t)nq + q];
    }
    return 0;
}

// row selection (topk.hip) as emulated kernel source: the k best of each row, canonical order
extern "C" int emu_row_select(const float* vals, int64_t nrows, int64_t n, int k, int is_l2, int64_t* out_keys, float* out_d) {
    int64_t kp = 2;
    while (kp < k) {
        kp <<= 1;
    }
    std::vector<unsigned long long> scratch((size_t)k > row_select_lds_max_k() ? (size_t)nrows * kp : 1);
    return launch_row_select(vals, nrows, n, k, is_l2 != 0, out_keys, out_d, nullptr, nullptr, scratch.data()) == hipSuccess ? 0 : 1;
}

extern "C" int emu_merge_partials(const float* pd, const int64_t* pi, int64_t nq, int nslot, int k, int is_l2, float* out_d,
                                  int64_t* out_i) {
    return launch_merge_partials(pd, pi, nq, nslot, k, (int64_t)nslot * k, k, is_l2 != 0, out_d, out_i, nullptr, nullptr) ==
                           hipSuccess
                   ? 0
                   : 1;
}

// range.hip::pq_adc_dump_kernel as emulated kernel source: every exact ADC distance of every probed list.
// t2t [nq][256][M] = <q_m, cb[m][c]> and the c-major term-2 table are prepared here on the host (in the product they
// come from pq_query_table_kernel / pq_precomp_table_kernel, validated on hardware).
extern "C" int emu_pq_adc_dump(int64_t nlist, const int64_t* list_len, const int64_t* list_row_off, const uint8_t* codes, int M,
                               int d, int lut_mode, const float* precomp /* [nlist][M][256] or null */,
                               const float* cb /* [M][256][dsub] */, const float* centroids, const float* xq, int64_t nq,
                               int nprobe, const int64_t* keys, const float* cdis, int64_t ncol, float* dist) {
    const int dsub = d / M;
    std::vector<float> t2t((size_t)nq * 256 * M), precomp_t;
    for (int64_t q = 0; q < nq; q++) {
        for (int m = 0; m < M; m++) {
            for (int c = 0; c < 256; c++) {
                float t = 0.f;
                for (int i = 0; i < dsub; i++) {
                    t = ip_step(t, xq[q * d + m * dsub + i], cb[((size_t)m * 256 + c) * dsub + i]);
                }
                t2t[((size_t)q * 256 + c) * M + m] = t;
            }
        }
    }
    if (precomp != nullptr) {
        precomp_t.resize((size_t)nlist * 256 * M);
        for (int64_t l = 0; l < nlist; l++) {
            for (int m = 0; m < M; m++) {
                for (int c = 0; c < 256; c++) {
                    precomp_t[((size_t)l * 256 + c) * M + m] = precomp[((size_t)l * M + m) * 256 + c];
                }
            }
        }
    }
    PqDumpArgs a{};
    a.dist = dist;
    a.ncol = ncol;
    a.keys = keys;
    a.coarse_dis = cdis;
    a.nprobe = nprobe;
    a.nlist = nlist;
    a.list_len = list_len;
    a.list_row_off = list_row_off;
    a.codes = codes;
    a.M = M;
    a.d = d;
    a.lut_mode = lut_mode;
    a.t2t = t2t.data();
    a.precomp_t = precomp_t.empty() ? nullptr : precomp_t.data();
    a.cb = cb;
    a.centroids = centroids;
    a.queries = xq;
    return launch_pq_adc_dump(a, nq, lut_mode != PQ_LUT_IP, nullptr) == hipSuccess ? 0 : 1;
}
