// knowhere_amd/host/faiss_io.h -- reader / writer for the FAISS index byte format of the index
// kinds on the Search() path, so that an index serialised by the CPU nodes loads into the HIP node
// unchanged, and the other way round (SURVEY.md 8f rank 3 "same index bytes").
//
// Format sources (reference, /root/reference/thirdparty/faiss/faiss):
//   impl/index_write.cpp:100-111   index header  {int d; int64 ntotal; int64 dummy x2; bool is_trained;
//                                                  int metric_type; [float metric_arg if metric > 1]}
//   cppcontrib/knowhere/impl/index_write.cpp:80-103  Knowhere's variant of the same 16 dummy bytes:
//                                                  {bool is_cosine; u8 x3; u32; int64} (all zero otherwise)
//   impl/index_write.cpp:465-473   IVF header    {index header; size_t nlist; size_t nprobe;
//                                                  quantizer index; direct map}
//   impl/index_write.cpp:451-463   direct map    {char type; vector<int64> array; [hashtable pairs]}
//   impl/index_write.cpp:489-499   "IxF2"/"IxFI" {index header; xb as vector of 4-byte units}
//   cppcontrib/knowhere/impl/index_write.cpp:539-546  "IxF9" (IndexFlatCosine) {index header; xb; vector<float> L2 norms};
//                                                  read side also accepts "IxFI" + is_cosine byte + norms (index_read.cpp:813-831)
//   impl/index_write.cpp:738-744   "IwFl"        {IVF header; inverted lists}
//   impl/index_write.cpp:745-753   "IwSq"        {IVF header; ScalarQuantizer; size_t code_size;
//                                                  bool by_residual; inverted lists}
//   impl/index_write.cpp:799-808   "IwPQ"        {IVF header; bool by_residual; size_t code_size;
//                                                  ProductQuantizer; inverted lists}
//   impl/index_write.cpp:183-188   ProductQuantizer {size_t d, M, nbits; vector<float> centroids}
//   impl/index_write.cpp:262-269   ScalarQuantizer  {int qtype; int rangestat; float rangestat_arg;
//                                                  size_t d; size_t code_size; vector<float> trained}
//   impl/index_write.cpp:303-343   "ilar"        {size_t nlist; size_t code_size; u32 "full"|"sprs";
//                                                  vector<size_t> sizes; per non-empty list: codes, ids
//                                                  [, float norms for Knowhere's cosine IVF-Flat,
//                                                  cppcontrib/knowhere/impl/index_write.cpp:299-306]}
//   impl/index_write.cpp:849-858   "IxRF"        {index header; base index; refine index; float k_factor}
//   impl/index_write.cpp:692-699   "IxSQ"        {index header; scalar quantizer; codes} (a quantised refine index)
// vector<T> = {size_t n; T[n]}.  Everything little-endian, unaligned.
//
// Parse() followed by Write() reproduces the input bytes exactly (tests/test_faiss_io.py).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace knhip_host {

constexpr uint32_t FourCC(const char (&s)[5]) {
    return (uint32_t)(uint8_t)s[0] | ((uint32_t)(uint8_t)s[1] << 8) | ((uint32_t)(uint8_t)s[2] << 16) |
           ((uint32_t)(uint8_t)s[3] << 24);
}

struct FaissHeader {  // index header as it sits on the wire
    int32_t d = 0;
    int64_t ntotal = 0;
    uint8_t dummy[16] = {0};  // baseline faiss: two int64 (1 << 20); Knowhere: byte 0 = is_cosine
    bool is_trained = true;
    int32_t metric = 1;       // faiss::MetricType: 0 = inner product, 1 = L2
    float metric_arg = 0.f;   // on the wire only if metric > 1
    bool is_cosine() const { return dummy[0] == 1 && dummy[1] == 0 && dummy[2] == 0; }
};

struct FaissFlat {  // "IxF2" / "IxFI"
    uint32_t fourcc = 0;
    FaissHeader hdr;
    std::vector<float> xb;
};

struct FaissSQFlat {  // "IxSQ": faiss::IndexScalarQuantizer (impl/index_write.cpp:692-699, write_ScalarQuantizer :262-269)
    FaissHeader hdr;
    int32_t qtype = 0, rangestat = 0;  // QuantizerType: 0 QT_8bit, 4 QT_fp16, 7 QT_bf16; RangeStat 0 = RS_minmax
    float rangestat_arg = 0.f;
    uint64_t d = 0, code_size = 0;
    std::vector<float> trained;
    std::vector<uint8_t> codes;       // [ntotal][code_size]
};

struct FaissIndexData {
    // outer IndexRefine wrapper ("IxRF"), present when an IVF-PQ / IVF-SQ8 index was built with refine; the refine index
    // is flat fp32 (refine_index) or, for refine_type = fp16 / bf16 / sq8, a scalar quantizer store (refine_sq)
    bool has_refine = false;
    bool refine_is_sq = false;
    FaissHeader refine_hdr;
    FaissFlat refine_index;
    FaissSQFlat refine_sq;
    float k_factor = 1.f;

    uint32_t fourcc = 0;  // IwFl / IwSq / IwPQ / IxF2 / IxFI
    FaissHeader hdr;
    // flat ("IxF9" / cosine "IxFI": the L2 norms of the rows follow them)
    std::vector<float> xb;
    std::vector<float> flat_norms;
    // IVF header
    uint64_t nlist = 0, nprobe = 1;
    FaissFlat quantizer;
    int8_t direct_map_type = 0;
    std::vector<int64_t> direct_map_array;
    std::vector<int64_t> direct_map_hash;  // flattened (key, value) pairs when type == Hashtable (2)
    // IwPQ / IwSq
    bool by_residual = true;
    uint64_t code_size = 0;
    uint64_t pq_d = 0, pq_M = 0, pq_nbits = 8;
    std::vector<float> pq_centroids;  // [M][ksub][dsub]
    int32_t sq_qtype = 0, sq_rangestat = 0;
    float sq_rangestat_arg = 0.f;
    uint64_t sq_d = 0, sq_code_size = 0;
    std::vector<float> sq_trained;    // QT_8bit: vmin[d] then vdiff[d]
    // inverted lists
    bool lists_sparse = false;        // which of the two size encodings was on the wire
    bool with_norm = false;           // Knowhere cosine IVF-Flat: a float norm per entry after the ids
    std::vector<std::vector<uint8_t>> codes;
    std::vector<std::vector<int64_t>> ids;
    std::vector<std::vector<float>> norms;
};

// Both return false and set *err on malformed / unsupported input; nothing throws.
bool ParseFaissIndex(const uint8_t* data, size_t size, FaissIndexData* out, std::string* err);
bool WriteFaissIndex(const FaissIndexData& in, std::vector<uint8_t>* out, std::string* err);

}  // namespace knhip_host

// C entry point used by the tests: parse + re-emit.  Returns the number of bytes written (the size
// needed if cap is too small), or -1 with a message in err (if err_cap > 0).
extern "C" int64_t knhip_host_faiss_roundtrip(const uint8_t* data, int64_t size, uint8_t* out, int64_t cap,
                                              char* err, int64_t err_cap);
// C entry point used by the tests: summary of a parsed blob.
//   info[0..9] = fourcc, d, ntotal, metric, nlist, code_size, pq_M, has_refine, is_cosine, sum of list sizes
extern "C" int knhip_host_faiss_info(const uint8_t* data, int64_t size, int64_t* info, char* err, int64_t err_cap);
