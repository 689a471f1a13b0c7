// knowhere_amd/host/faiss_io.cc -- see faiss_io.h for the format and its reference sources.
#include "faiss_io.h"

#include <cstring>
#include <stdexcept>

namespace knhip_host {
namespace {

struct Reader {
    const uint8_t* p;
    const uint8_t* end;
    void raw(void* dst, size_t n) {
        if ((size_t)(end - p) < n) throw std::runtime_error("truncated index blob");
        std::memcpy(dst, p, n);
        p += n;
    }
    template <typename T>
    T one() {
        T v;
        raw(&v, sizeof(T));
        return v;
    }
    template <typename T>
    void vec(std::vector<T>& v, size_t unit_scale = 1) {
        const uint64_t n = one<uint64_t>();
        if (n > (uint64_t)(end - p)) throw std::runtime_error("vector length exceeds blob");
        const uint64_t count = n * unit_scale;
        if (count * sizeof(T) > (uint64_t)(end - p)) throw std::runtime_error("vector length exceeds blob");
        v.resize(count);
        raw(v.data(), count * sizeof(T));
    }
};

struct Writer {
    std::vector<uint8_t>* o;
    void raw(const void* src, size_t n) { o->insert(o->end(), (const uint8_t*)src, (const uint8_t*)src + n); }
    template <typename T>
    void one(T v) {
        raw(&v, sizeof(T));
    }
    template <typename T>
    void vec(const std::vector<T>& v, size_t unit_scale = 1) {
        one<uint64_t>(v.size() / unit_scale);
        raw(v.data(), v.size() * sizeof(T));
    }
};

void ReadHeader(Reader& r, FaissHeader& h) {
    h.d = r.one<int32_t>();
    h.ntotal = r.one<int64_t>();
    r.raw(h.dummy, 16);
    h.is_trained = r.one<uint8_t>() != 0;
    h.metric = r.one<int32_t>();
    if (h.metric > 1) h.metric_arg = r.one<float>();
    if (h.d <= 0 || h.ntotal < 0) throw std::runtime_error("bad index header");
}

void WriteHeader(Writer& w, const FaissHeader& h) {
    w.one<int32_t>(h.d);
    w.one<int64_t>(h.ntotal);
    w.raw(h.dummy, 16);
    w.one<uint8_t>(h.is_trained ? 1 : 0);
    w.one<int32_t>(h.metric);
    if (h.metric > 1) w.one<float>(h.metric_arg);
}

bool IsFlat(uint32_t f) { return f == FourCC("IxF2") || f == FourCC("IxFI"); }

void ReadFlatBody(Reader& r, FaissFlat& f) {
    ReadHeader(r, f.hdr);
    r.vec(f.xb);  // READXBVECTOR: count of 4-byte units == count of floats
    if ((int64_t)f.xb.size() != f.hdr.ntotal * f.hdr.d) throw std::runtime_error("flat index size mismatch");
}

void WriteFlat(Writer& w, const FaissFlat& f) {
    w.one<uint32_t>(f.fourcc);
    WriteHeader(w, f.hdr);
    w.vec(f.xb);
}

void ReadInvertedLists(Reader& r, FaissIndexData& x) {
    const uint32_t h = r.one<uint32_t>();
    if (h != FourCC("ilar")) throw std::runtime_error("unsupported inverted-list container (only ArrayInvertedLists)");
    const uint64_t nlist = r.one<uint64_t>();
    const uint64_t cs = r.one<uint64_t>();
    if (nlist != x.nlist) throw std::runtime_error("inverted lists: nlist mismatch");
    if (nlist > (uint64_t)(r.end - r.p)) throw std::runtime_error("inverted lists: nlist exceeds blob");
    if (x.code_size == 0) x.code_size = cs;
    if (cs != x.code_size) throw std::runtime_error("inverted lists: code_size mismatch");
    const uint32_t lt = r.one<uint32_t>();
    std::vector<uint64_t> sizes;
    r.vec(sizes);
    std::vector<uint64_t> n(nlist, 0);
    if (lt == FourCC("full")) {
        if (sizes.size() != nlist) throw std::runtime_error("inverted lists: bad size table");
        n = sizes;
        x.lists_sparse = false;
    } else if (lt == FourCC("sprs")) {
        if (sizes.size() % 2) throw std::runtime_error("inverted lists: bad sparse size table");
        for (size_t i = 0; i < sizes.size(); i += 2) {
            if (sizes[i] >= nlist) throw std::runtime_error("inverted lists: list number out of range");
            n[sizes[i]] = sizes[i + 1];
        }
        x.lists_sparse = true;
    } else {
        throw std::runtime_error("inverted lists: unknown list type");
    }
    x.codes.assign(nlist, {});
    x.ids.assign(nlist, {});
    x.norms.assign(x.with_norm ? nlist : 0, {});
    for (uint64_t l = 0; l < nlist; l++) {
        if (n[l] == 0) continue;
        if (cs == 0 || n[l] > (uint64_t)(r.end - r.p) / cs) throw std::runtime_error("inverted lists: list length exceeds blob");
        x.codes[l].resize(n[l] * cs);
        r.raw(x.codes[l].data(), n[l] * cs);
        x.ids[l].resize(n[l]);
        r.raw(x.ids[l].data(), n[l] * 8);
        if (x.with_norm) {
            x.norms[l].resize(n[l]);
            r.raw(x.norms[l].data(), n[l] * 4);
        }
    }
}

void WriteInvertedLists(Writer& w, const FaissIndexData& x) {
    w.one<uint32_t>(FourCC("ilar"));
    w.one<uint64_t>(x.nlist);
    w.one<uint64_t>(x.code_size);
    std::vector<uint64_t> sizes;
    if (!x.lists_sparse) {
        w.one<uint32_t>(FourCC("full"));
        for (uint64_t l = 0; l < x.nlist; l++) sizes.push_back(x.ids[l].size());
    } else {
        w.one<uint32_t>(FourCC("sprs"));
        for (uint64_t l = 0; l < x.nlist; l++)
            if (!x.ids[l].empty()) {
                sizes.push_back(l);
                sizes.push_back(x.ids[l].size());
            }
    }
    w.vec(sizes);
    for (uint64_t l = 0; l < x.nlist; l++) {
        if (x.ids[l].empty()) continue;
        w.raw(x.codes[l].data(), x.codes[l].size());
        w.raw(x.ids[l].data(), x.ids[l].size() * 8);
        if (x.with_norm) w.raw(x.norms[l].data(), x.norms[l].size() * 4);
    }
}

void ReadIvfHeader(Reader& r, FaissIndexData& x) {
    ReadHeader(r, x.hdr);
    x.nlist = r.one<uint64_t>();
    x.nprobe = r.one<uint64_t>();
    if (x.nlist == 0 || x.nlist > (uint64_t)(r.end - r.p)) throw std::runtime_error("bad nlist");
    x.quantizer.fourcc = r.one<uint32_t>();
    if (!IsFlat(x.quantizer.fourcc)) throw std::runtime_error("coarse quantizer is not a flat index");
    ReadFlatBody(r, x.quantizer);
    if (x.quantizer.hdr.d != x.hdr.d || (uint64_t)x.quantizer.hdr.ntotal != x.nlist)
        throw std::runtime_error("coarse quantizer shape mismatch");
    x.direct_map_type = r.one<int8_t>();
    r.vec(x.direct_map_array);
    if (x.direct_map_type == 2) r.vec(x.direct_map_hash, 2);
}

void WriteIvfHeader(Writer& w, const FaissIndexData& x) {
    WriteHeader(w, x.hdr);
    w.one<uint64_t>(x.nlist);
    w.one<uint64_t>(x.nprobe);
    WriteFlat(w, x.quantizer);
    w.one<int8_t>(x.direct_map_type);
    w.vec(x.direct_map_array);
    if (x.direct_map_type == 2) w.vec(x.direct_map_hash, 2);
}

void ReadBody(Reader& r, uint32_t h, FaissIndexData& x) {
    x.fourcc = h;
    if (IsFlat(h) || h == FourCC("IxF9")) {
        FaissFlat f;
        ReadFlatBody(r, f);
        x.hdr = f.hdr;
        x.xb = std::move(f.xb);
        // Knowhere's cosine flat index: "IxF9", or "IxFI" with the is_cosine header byte, carries the L2 norms of
        // the (raw) rows behind them (cppcontrib/knowhere/impl/index_write.cpp:539-546, index_read.cpp:784-831)
        if (h == FourCC("IxF9") || x.hdr.is_cosine()) {
            r.vec(x.flat_norms);
            if ((int64_t)x.flat_norms.size() != x.hdr.ntotal) throw std::runtime_error("flat cosine norms size mismatch");
        }
    } else if (h == FourCC("IwFl")) {
        ReadIvfHeader(r, x);
        x.code_size = (uint64_t)x.hdr.d * 4;
        x.with_norm = x.hdr.is_cosine();
        ReadInvertedLists(r, x);
    } else if (h == FourCC("IwSq")) {
        ReadIvfHeader(r, x);
        x.sq_qtype = r.one<int32_t>();
        x.sq_rangestat = r.one<int32_t>();
        x.sq_rangestat_arg = r.one<float>();
        x.sq_d = r.one<uint64_t>();
        x.sq_code_size = r.one<uint64_t>();
        r.vec(x.sq_trained);
        x.code_size = r.one<uint64_t>();
        x.by_residual = r.one<uint8_t>() != 0;
        ReadInvertedLists(r, x);
    } else if (h == FourCC("IwPQ")) {
        ReadIvfHeader(r, x);
        x.by_residual = r.one<uint8_t>() != 0;
        x.code_size = r.one<uint64_t>();
        x.pq_d = r.one<uint64_t>();
        x.pq_M = r.one<uint64_t>();
        x.pq_nbits = r.one<uint64_t>();
        r.vec(x.pq_centroids);
        if (x.pq_d != (uint64_t)x.hdr.d || x.pq_M == 0 || x.pq_d % x.pq_M || x.pq_nbits > 16 ||
            x.pq_centroids.size() != (x.pq_d << x.pq_nbits))
            throw std::runtime_error("product quantizer shape mismatch");
        ReadInvertedLists(r, x);
    } else {
        char cc[5] = {(char)(h & 255), (char)((h >> 8) & 255), (char)((h >> 16) & 255), (char)(h >> 24), 0};
        throw std::runtime_error(std::string("index type '") + cc + "' is not on the HIP Search() path");
    }
}

void WriteBody(Writer& w, const FaissIndexData& x) {
    w.one<uint32_t>(x.fourcc);
    if (IsFlat(x.fourcc) || x.fourcc == FourCC("IxF9")) {
        WriteHeader(w, x.hdr);
        w.vec(x.xb);
        if (x.fourcc == FourCC("IxF9") || x.hdr.is_cosine()) w.vec(x.flat_norms);
    } else if (x.fourcc == FourCC("IwFl")) {
        WriteIvfHeader(w, x);
        WriteInvertedLists(w, x);
    } else if (x.fourcc == FourCC("IwSq")) {
        WriteIvfHeader(w, x);
        w.one<int32_t>(x.sq_qtype);
        w.one<int32_t>(x.sq_rangestat);
        w.one<float>(x.sq_rangestat_arg);
        w.one<uint64_t>(x.sq_d);
        w.one<uint64_t>(x.sq_code_size);
        w.vec(x.sq_trained);
        w.one<uint64_t>(x.code_size);
        w.one<uint8_t>(x.by_residual ? 1 : 0);
        WriteInvertedLists(w, x);
    } else if (x.fourcc == FourCC("IwPQ")) {
        WriteIvfHeader(w, x);
        w.one<uint8_t>(x.by_residual ? 1 : 0);
        w.one<uint64_t>(x.code_size);
        w.one<uint64_t>(x.pq_d);
        w.one<uint64_t>(x.pq_M);
        w.one<uint64_t>(x.pq_nbits);
        w.vec(x.pq_centroids);
        WriteInvertedLists(w, x);
    } else {
        throw std::runtime_error("cannot write this index type");
    }
}

}  // namespace

bool ParseFaissIndex(const uint8_t* data, size_t size, FaissIndexData* out, std::string* err) {
    try {
        *out = FaissIndexData();
        Reader r{data, data + size};
        uint32_t h = r.one<uint32_t>();
        if (h == FourCC("IxRF")) {
            out->has_refine = true;
            ReadHeader(r, out->refine_hdr);
            ReadBody(r, r.one<uint32_t>(), *out);
            const uint32_t rcc = r.one<uint32_t>();
            if (rcc == FourCC("IxSQ")) {
                FaissSQFlat& q = out->refine_sq;
                out->refine_is_sq = true;
                ReadHeader(r, q.hdr);
                q.qtype = r.one<int32_t>();
                q.rangestat = r.one<int32_t>();
                q.rangestat_arg = r.one<float>();
                q.d = r.one<uint64_t>();
                q.code_size = r.one<uint64_t>();
                r.vec(q.trained);
                r.vec(q.codes);
                // QT_8bit 0, QT_4bit_uniform 3, QT_fp16 4, QT_6bit 6, QT_bf16 7, QT_8bit_direct_signed 8 (impl/ScalarQuantizer.h:27-40)
                const uint64_t want = (q.qtype == 0 || q.qtype == 8)   ? q.d
                                      : (q.qtype == 4 || q.qtype == 7) ? 2 * q.d
                                      : q.qtype == 6                   ? (q.d * 6 + 7) / 8
                                      : q.qtype == 3                   ? (q.d * 4 + 7) / 8
                                                                       : 0;
                if (want == 0)
                    throw std::runtime_error("refine scalar quantizer type is not fp16 / bf16 / sq8 / sq6 / int8 / sq4u");
                const uint64_t ntrained = (q.qtype == 0 || q.qtype == 6) ? 2 * q.d : q.qtype == 3 ? 2 : 0;
                if (q.d != (uint64_t)q.hdr.d || q.code_size != want || q.codes.size() != (uint64_t)q.hdr.ntotal * want ||
                    q.trained.size() != ntrained)
                    throw std::runtime_error("refine scalar quantizer shape mismatch");
            } else {
                out->refine_index.fourcc = rcc;
                if (!IsFlat(rcc)) throw std::runtime_error("refine index is neither flat fp32 nor a scalar quantizer store");
                ReadFlatBody(r, out->refine_index);
            }
            out->k_factor = r.one<float>();
        } else {
            ReadBody(r, h, *out);
        }
        if (r.p != r.end) throw std::runtime_error("trailing bytes after the index");
        return true;
    } catch (const std::exception& e) {
        if (err) *err = e.what();
        return false;
    }
}

bool WriteFaissIndex(const FaissIndexData& in, std::vector<uint8_t>* out, std::string* err) {
    try {
        out->clear();
        Writer w{out};
        if (in.has_refine) {
            w.one<uint32_t>(FourCC("IxRF"));
            WriteHeader(w, in.refine_hdr);
            WriteBody(w, in);
            if (in.refine_is_sq) {
                const FaissSQFlat& q = in.refine_sq;
                w.one<uint32_t>(FourCC("IxSQ"));
                WriteHeader(w, q.hdr);
                w.one<int32_t>(q.qtype);
                w.one<int32_t>(q.rangestat);
                w.one<float>(q.rangestat_arg);
                w.one<uint64_t>(q.d);
                w.one<uint64_t>(q.code_size);
                w.vec(q.trained);
                w.vec(q.codes);
            } else {
                WriteFlat(w, in.refine_index);
            }
            w.one<float>(in.k_factor);
        } else {
            WriteBody(w, in);
        }
        return true;
    } catch (const std::exception& e) {
        if (err) *err = e.what();
        return false;
    }
}

}  // namespace knhip_host

namespace {
void SetErr(char* err, int64_t cap, const std::string& s) {
    if (err && cap > 0) {
        std::strncpy(err, s.c_str(), (size_t)cap - 1);
        err[cap - 1] = 0;
    }
}
}  // namespace

extern "C" int64_t knhip_host_faiss_roundtrip(const uint8_t* data, int64_t size, uint8_t* out, int64_t cap,
                                              char* err, int64_t err_cap) {
    knhip_host::FaissIndexData x;
    std::string e;
    if (!knhip_host::ParseFaissIndex(data, (size_t)size, &x, &e)) {
        SetErr(err, err_cap, e);
        return -1;
    }
    std::vector<uint8_t> buf;
    if (!knhip_host::WriteFaissIndex(x, &buf, &e)) {
        SetErr(err, err_cap, e);
        return -1;
    }
    if ((int64_t)buf.size() <= cap) std::memcpy(out, buf.data(), buf.size());
    return (int64_t)buf.size();
}

extern "C" int knhip_host_faiss_info(const uint8_t* data, int64_t size, int64_t* info, char* err, int64_t err_cap) {
    knhip_host::FaissIndexData x;
    std::string e;
    if (!knhip_host::ParseFaissIndex(data, (size_t)size, &x, &e)) {
        SetErr(err, err_cap, e);
        return -1;
    }
    int64_t total = 0;
    for (auto& l : x.ids) total += (int64_t)l.size();
    info[0] = x.fourcc; info[1] = x.hdr.d; info[2] = x.hdr.ntotal; info[3] = x.hdr.metric; info[4] = (int64_t)x.nlist;
    info[5] = (int64_t)x.code_size; info[6] = (int64_t)x.pq_M; info[7] = x.has_refine; info[8] = x.hdr.is_cosine();
    info[9] = total;
    return 0;
}
