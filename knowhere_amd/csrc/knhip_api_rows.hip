// knowhere_amd/csrc/knhip_api_rows.hip -- the quantised refine store behind the C ABI (include/knhip.h knhip_rows_*: Knowhere's
// refine_type = fp16 / bf16 / sq8 / sq6 / int8 / sq4u, the refine index as a faiss::IndexScalarQuantizer, reference
// src/index/refine/refine_utils.cc:99-160): create / train / add / read back.  The re-rank over it: refine.hip through
// knhip_search_refine_rows / knhip_refine_rows_device (knhip_api.hip).
#include "knhip_internal.h"

extern "C" {

int knhip_rows_create(int32_t device, int32_t dim, int32_t row_type, knhip_rows** out) {
    if (!out || dim <= 0 || row_type < KNHIP_ROWS_FP16 || row_type > KNHIP_ROWS_SQ4U) {
        return fail(KNHIP_ERR_INVALID_ARGS, "rows_create: dim > 0 and row type fp16 / bf16 / sq8 / sq6 / int8 / sq4u");
    }
    if (device < 0 || device >= knhip_device_count()) {
        return fail(KNHIP_ERR_HIP_RUNTIME, "rows_create: no such HIP device");
    }
    auto* r = new knhip_rows();
    r->device = device;
    r->d = dim;
    r->row_type = row_type;
    r->trained = !r->ranged(); // (fp16 / bf16 / int8 have nothing to train)
    *out = r;
    return KNHIP_OK;
}

void knhip_rows_destroy(knhip_rows* r) {
    if (r) {
        DeviceGuard g(r->device);
        delete r;
    }
}

int64_t knhip_rows_count(const knhip_rows* r) { return r ? r->n : 0; }
int64_t knhip_rows_code_size(const knhip_rows* r) { return r ? r->code_size() : 0; }
int64_t knhip_rows_device_bytes(const knhip_rows* r) { return r ? (int64_t)(r->codes.bytes + r->sq.bytes) : 0; }

int knhip_rows_set_trained(knhip_rows* r, const float* vmin, const float* vdiff) {
    if (!r || !r->ranged() || !vmin || !vdiff) {
        return fail(KNHIP_ERR_INVALID_ARGS, "rows_set_trained: an sq8 / sq6 / sq4u store and two arrays of dim (sq4u: 1) floats");
    }
    DeviceGuard g(r->device);
    std::lock_guard<std::mutex> lk(r->mu);
    const size_t nr = (size_t)r->nrange();
    HIP_TRY(r->sq.reserve(2 * nr * sizeof(float)));
    HIP_TRY(hipMemcpy(r->sq.p, vmin, nr * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(r->sq.as<float>() + nr, vdiff, nr * sizeof(float), hipMemcpyHostToDevice));
    r->trained = true;
    return KNHIP_OK;
}

int knhip_rows_get_trained(const knhip_rows* r, float* vmin, float* vdiff) {
    if (!r || !r->ranged() || !r->trained || !vmin || !vdiff) {
        return fail(KNHIP_ERR_NOT_TRAINED, "rows_get_trained: a trained ranged store (sq8 / sq6 / sq4u)");
    }
    DeviceGuard g(r->device);
    const size_t nr = (size_t)r->nrange();
    HIP_TRY(hipMemcpy(vmin, r->sq.p, nr * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(vdiff, r->sq.as<float>() + nr, nr * sizeof(float), hipMemcpyDeviceToHost));
    return KNHIP_OK;
}

// ScalarQuantizer::train, QT_8bit / QT_6bit, RS_minmax with rangestat_arg 0 (impl/ScalarQuantizer.cpp train_NonUniform): vmin = column
// minimum, vdiff = column maximum - vmin over ALL n rows (no sub-sampling for RS_minmax)
int knhip_rows_train(knhip_rows* r, int64_t n, const float* x) {
    if (!r || n < 0 || (n > 0 && !x)) {
        return fail(KNHIP_ERR_INVALID_ARGS, "rows_train: bad arguments");
    }
    if (!r->ranged()) {
        return KNHIP_OK;
    }
    if (r->row_type == KNHIP_ROWS_SQ4U) {
        return knhip_rows_train_uniform(r, n, x, 0, 0.f); // (RS_minmax, argument 0: the ScalarQuantizer defaults)
    }
    if (n == 0) {
        return fail(KNHIP_ERR_INVALID_ARGS, "rows_train: no training rows");
    }
    DeviceGuard g(r->device);
    const int d = r->d;
    std::vector<float> lo((size_t)d, INFINITY), hi((size_t)d, -INFINITY), a((size_t)d), b((size_t)d);
    DevBuf dx, mm;
    HIP_TRY(mm.alloc((size_t)2 * d * sizeof(float)));
    const int64_t step = std::max<int64_t>(1, ((int64_t)1 << 30) / ((int64_t)d * 4));
    for (int64_t i0 = 0; i0 < n; i0 += step) { // (min / max are order independent: slices of at most 1 GiB)
        const int64_t m = std::min(step, n - i0);
        if (int rc = upload(dx, x + i0 * d, (size_t)m * d * sizeof(float))) return rc;
        HIP_TRY(launch_col_minmax(dx.as<float>(), m, d, mm.as<float>(), mm.as<float>() + d, nullptr));
        HIP_TRY(hipMemcpy(a.data(), mm.p, (size_t)d * sizeof(float), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(b.data(), mm.as<float>() + d, (size_t)d * sizeof(float), hipMemcpyDeviceToHost));
        for (int j = 0; j < d; j++) {
            lo[(size_t)j] = std::min(lo[(size_t)j], a[(size_t)j]);
            hi[(size_t)j] = std::max(hi[(size_t)j], b[(size_t)j]);
        }
    }
    for (int j = 0; j < d; j++) {
        hi[(size_t)j] = hi[(size_t)j] - lo[(size_t)j];
    }
    return knhip_rows_set_trained(r, lo.data(), hi.data());
}

// ScalarQuantizer::train for QT_4bit_uniform: train_Uniform over the n * d values (impl/scalar_quantizer/training.cpp:209-332).
// RS_minmax: min / max widened by arg * (max - min).  RS_quantiles: o = (idx_t)(arg * N) -- a FLOAT product, as the
// reference forms it --, vmin = the o-th smallest value, vmax = the (N - 1 - o)-th: a radix select over the order-
// preserving keys, eight bits per pass, the host slices re-uploaded per pass (training runs once per index).
int knhip_rows_train_uniform(knhip_rows* r, int64_t n, const float* x, int32_t rangestat, float rangestat_arg) {
    if (!r || r->row_type != KNHIP_ROWS_SQ4U || n <= 0 || !x || (rangestat != 0 && rangestat != 2)) {
        return fail(KNHIP_ERR_INVALID_ARGS, "rows_train_uniform: an sq4u store, training rows, rangestat 0 (minmax) or 2 (quantiles)");
    }
    DeviceGuard g(r->device);
    const int d = r->d;
    const int64_t N = n * (int64_t)d;
    const int64_t step = (int64_t)1 << 28; // values per slice (1 GiB)
    float vmin = 0.f, vmax = 0.f;
    DevBuf dx;
    if (rangestat == 0) {
        DevBuf mm;
        HIP_TRY(mm.alloc((size_t)2 * d * sizeof(float)));
        std::vector<float> ab((size_t)2 * d);
        vmin = INFINITY;
        vmax = -INFINITY;
        const int64_t rstep = std::max<int64_t>(1, step / d);
        for (int64_t r0 = 0; r0 < n; r0 += rstep) { // (column extrema per slice of rows, reduced on the host: order independent)
            const int64_t m = std::min(rstep, n - r0);
            if (int rc = upload(dx, x + r0 * d, (size_t)m * d * sizeof(float))) return rc;
            HIP_TRY(launch_col_minmax(dx.as<float>(), m, d, mm.as<float>(), mm.as<float>() + d, nullptr));
            HIP_TRY(hipMemcpy(ab.data(), mm.p, (size_t)2 * d * sizeof(float), hipMemcpyDeviceToHost));
            for (int j = 0; j < d; j++) {
                vmin = std::min(vmin, ab[(size_t)j]);
                vmax = std::max(vmax, ab[(size_t)d + j]);
            }
        }
        const float vexp = (vmax - vmin) * rangestat_arg;
        vmin -= vexp;
        vmax += vexp;
    } else {
        int64_t o = static_cast<int64_t>(rangestat_arg * N); // (float * idx_t: the count is converted to float)
        if (o < 0) o = 0;
        if (o > N - o) o = N / 2;
        int64_t rank[2] = {o, N - 1 - o}; // 0-based ranks still to find inside the current prefixes
        uint32_t prefix[2] = {0u, 0u}, mask = 0u;
        DevBuf dh;
        HIP_TRY(dh.alloc(512 * sizeof(unsigned long long)));
        std::vector<unsigned long long> h(512);
        const bool resident = N <= step;
        if (resident) {
            if (int rc = upload(dx, x, (size_t)N * sizeof(float))) return rc;
        }
        for (int shift = 24; shift >= 0; shift -= 8) {
            HIP_TRY(hipMemset(dh.p, 0, 512 * sizeof(unsigned long long)));
            for (int64_t i0 = 0; i0 < N; i0 += step) {
                const int64_t m = std::min(step, N - i0);
                if (!resident) {
                    if (int rc = upload(dx, x + i0, (size_t)m * sizeof(float))) return rc;
                }
                HIP_TRY(launch_rows_key_hist(dx.as<float>(), m, mask, prefix[0], prefix[1], shift,
                                             dh.as<unsigned long long>(), nullptr));
                HIP_TRY(hipDeviceSynchronize());
            }
            HIP_TRY(hipMemcpy(h.data(), dh.p, 512 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            for (int w = 0; w < 2; w++) {
                int b = 0;
                int64_t left = rank[w];
                while (b < 255 && left >= (int64_t)h[(size_t)w * 256 + b]) {
                    left -= (int64_t)h[(size_t)w * 256 + b];
                    b++;
                }
                rank[w] = left;
                prefix[w] |= (uint32_t)b << shift;
            }
            mask |= 0xffu << shift;
        }
        auto unkey = [](uint32_t key) {
            const uint32_t bits = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key;
            float f;
            std::memcpy(&f, &bits, 4);
            return f;
        };
        vmin = unkey(prefix[0]);
        vmax = unkey(prefix[1]);
    }
    const float vdiff = vmax - vmin;
    return knhip_rows_set_trained(r, &vmin, &vdiff);
}

static int rows_append(knhip_rows* r, int64_t n, const void* d_new_codes) {
    const size_t cs = (size_t)r->code_size();
    const size_t need = (size_t)(r->n + n) * cs;
    if (need > r->codes.bytes) { // grow by half: repeated Adds copy the store O(log) times (a search in flight must not
        DevBuf all;              // race an Add: the node's reader / writer lock, as for every other index state)
        HIP_TRY(all.alloc(std::max(need, r->codes.bytes + r->codes.bytes / 2)));
        if (r->n) {
            HIP_TRY(hipMemcpy(all.p, r->codes.p, (size_t)r->n * cs, hipMemcpyDeviceToDevice));
        }
        std::swap(r->codes.p, all.p);
        std::swap(r->codes.bytes, all.bytes);
    }
    HIP_TRY(hipMemcpy(static_cast<char*>(r->codes.p) + (size_t)r->n * cs, d_new_codes, (size_t)n * cs, hipMemcpyDeviceToDevice));
    r->n += n;
    return KNHIP_OK;
}

// encode (ScalarQuantizer::compute_codes) and append: row r of the store is vector id r
int knhip_rows_add(knhip_rows* r, int64_t n, const float* x) {
    if (!r || n < 0 || (n > 0 && !x)) {
        return fail(KNHIP_ERR_INVALID_ARGS, "rows_add: bad arguments");
    }
    if (!r->trained) {
        return fail(KNHIP_ERR_NOT_TRAINED, "rows_add: the ranges are not trained");
    }
    if (n == 0) {
        return KNHIP_OK;
    }
    DeviceGuard g(r->device);
    std::lock_guard<std::mutex> lk(r->mu);
    const int d = r->d;
    const int64_t step = std::max<int64_t>(1, ((int64_t)1 << 30) / ((int64_t)d * 4));
    for (int64_t i0 = 0; i0 < n; i0 += step) {
        const int64_t m = std::min(step, n - i0);
        DevBuf dx, dc;
        if (int rc = upload(dx, x + i0 * d, (size_t)m * d * sizeof(float))) return rc;
        HIP_TRY(dc.alloc((size_t)m * r->code_size()));
        if (r->row_type == KNHIP_ROWS_SQ8) {
            HIP_TRY(launch_sq8_encode(dx.as<float>(), m, d, r->sq.as<float>(), dc.as<uint8_t>(), nullptr));
        } else if (r->row_type == KNHIP_ROWS_SQ6) {
            HIP_TRY(launch_rows_encode6(dx.as<float>(), m, d, r->sq.as<float>(), dc.as<uint8_t>(), nullptr));
        } else if (r->row_type == KNHIP_ROWS_INT8) {
            HIP_TRY(launch_rows_encode_i8(dx.as<float>(), m * d, dc.as<uint8_t>(), nullptr));
        } else if (r->row_type == KNHIP_ROWS_SQ4U) {
            HIP_TRY(launch_rows_encode4u(dx.as<float>(), m, d, r->sq.as<float>(), dc.as<uint8_t>(), nullptr));
        } else {
            HIP_TRY(launch_rows_encode16(dx.as<float>(), m * d, r->row_type == KNHIP_ROWS_BF16, dc.as<uint16_t>(), nullptr));
        }
        HIP_TRY(hipDeviceSynchronize());
        if (int rc = rows_append(r, m, dc.p)) return rc;
    }
    return KNHIP_OK;
}

int knhip_rows_add_codes(knhip_rows* r, int64_t n, const uint8_t* codes) {
    if (!r || n < 0 || (n > 0 && !codes)) {
        return fail(KNHIP_ERR_INVALID_ARGS, "rows_add_codes: bad arguments");
    }
    if (n == 0) {
        return KNHIP_OK;
    }
    DeviceGuard g(r->device);
    std::lock_guard<std::mutex> lk(r->mu);
    DevBuf dc;
    if (int rc = upload(dc, codes, (size_t)n * r->code_size())) return rc;
    return rows_append(r, n, dc.p);
}

int knhip_rows_get_codes(const knhip_rows* r, uint8_t* out) {
    if (!r || !out) {
        return fail(KNHIP_ERR_INVALID_ARGS, "rows_get_codes: bad arguments");
    }
    DeviceGuard g(r->device);
    if (r->n) {
        HIP_TRY(hipMemcpy(out, r->codes.p, (size_t)r->n * r->code_size(), hipMemcpyDeviceToHost));
    }
    return KNHIP_OK;
}
} // extern "C"
