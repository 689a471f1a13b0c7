// knowhere_amd/host/node_capi.cc -- a C view of the IndexNode (IndexFactory::Create / Index::Build /
// Search / Serialize / Deserialize), so that non-C++ hosts and the Python tests can drive the plugin
// exactly as Knowhere callers do.  Config is passed as "key=value;key=value" with the reference's key
// names (include/knowhere/comp/index_param.h:100-180): integers, true/false, or strings.
#include "hip_index_node.h"

#include <cstdlib>
#include <cstring>

using namespace knowhere;

extern "C" void knhip_host_normalize_rows(float* x, int64_t n, int64_t d, float* norms);  // hip_index_node.cc
extern "C" int32_t knhip_host_last_placement(int32_t* out, int32_t cap);                  // hip_index_node.cc

namespace {
thread_local std::string g_err;

Json ParseConfig(const char* s) {
    Json j;
    std::string str = s ? s : "";
    size_t pos = 0;
    while (pos < str.size()) {
        size_t end = str.find(';', pos);
        if (end == std::string::npos) end = str.size();
        const std::string kv = str.substr(pos, end - pos);
        pos = end + 1;
        const size_t eq = kv.find('=');
        if (eq == std::string::npos) continue;
        const std::string k = kv.substr(0, eq), v = kv.substr(eq + 1);
        char* e = nullptr;
        const long long iv = std::strtoll(v.c_str(), &e, 10);
        if (v == "true" || v == "false") {
            j[k] = (v == "true");
        } else if (!v.empty() && e && *e == 0) {
            j[k] = (int64_t)iv;
        } else {
            char* e2 = nullptr;
            const double dv = std::strtod(v.c_str(), &e2);
            if (!v.empty() && e2 && *e2 == 0) j[k] = dv;
            else j[k] = v;
        }
    }
    return j;
}

struct Handle {
    Index<IndexNode> idx;
};
}  // namespace

extern "C" {

const char* knhip_node_last_error() { return g_err.c_str(); }

// IndexFactory::Instance().Create<fp32>(name, version); nullptr if the name is not registered
void* knhip_node_create(const char* name) {
    auto r = IndexFactory::Instance().Create<fp32>(name, Version::GetCurrentVersion().VersionNumber());
    if (!r.has_value()) {
        g_err = r.what();
        return nullptr;
    }
    return new Handle{r.value()};
}

void knhip_node_destroy(void* h) { delete static_cast<Handle*>(h); }

// Index::Build (= Train + Add); returns the Status value (0 = success)
int knhip_node_build(void* h, const float* x, int64_t rows, int64_t dim, const char* cfg) {
    auto ds = GenDataSet(rows, dim, x);
    return (int)static_cast<Handle*>(h)->idx.Build(ds, ParseConfig(cfg));
}

// Index::Search; ids/dist receive nq * k results.  bitset may be null.
int knhip_node_search(void* h, const float* q, int64_t nq, int64_t dim, const char* cfg, const uint8_t* bitset,
                      int64_t nbits, int64_t k, int64_t* ids, float* dist) {
    auto ds = GenDataSet(nq, dim, q);
    auto r = static_cast<Handle*>(h)->idx.Search(ds, ParseConfig(cfg), bitset ? BitsetView(bitset, (size_t)nbits) : BitsetView());
    if (!r.has_value()) {
        g_err = r.what();
        return (int)r.error();
    }
    std::memcpy(ids, r.value()->GetIds(), sizeof(int64_t) * nq * k);
    std::memcpy(dist, r.value()->GetDistance(), sizeof(float) * nq * k);
    return 0;
}

// Index::Serialize: copies the single blob of the BinarySet; returns its size (also when cap is too
// small) or -status.
int64_t knhip_node_serialize(void* h, uint8_t* out, int64_t cap) {
    BinarySet bs;
    auto& idx = static_cast<Handle*>(h)->idx;
    Status s = idx.Serialize(bs);
    if (s != Status::success) return -(int64_t)s;
    auto b = bs.GetByName(idx.Type());
    if (!b) return -(int64_t)Status::invalid_binary_set;
    if (b->size <= cap) std::memcpy(out, b->data.get(), (size_t)b->size);
    return b->size;
}

// Index::Deserialize from a BinarySet holding `data` under `key` (e.g. "IVF_PQ" for a CPU-built index)
int knhip_node_deserialize(void* h, const char* key, const uint8_t* data, int64_t size, const char* cfg) {
    BinarySet bs;
    std::shared_ptr<uint8_t[]> copy(new uint8_t[size]);
    std::memcpy(copy.get(), data, (size_t)size);
    bs.Append(key, copy, size);
    return (int)static_cast<Handle*>(h)->idx.Deserialize(bs, ParseConfig(cfg));
}

// Index::RangeSearch: lims[nq + 1]; *ids / *dist are malloc'ed copies (release with free())
int knhip_node_range_search(void* h, const float* q, int64_t nq, int64_t dim, const char* cfg, const uint8_t* bitset,
                            int64_t nbits, int64_t* lims, int64_t** ids, float** dist) {
    auto ds = GenDataSet(nq, dim, q);
    auto r = static_cast<Handle*>(h)->idx.RangeSearch(ds, ParseConfig(cfg),
                                                      bitset ? BitsetView(bitset, (size_t)nbits) : BitsetView());
    if (!r.has_value()) {
        g_err = r.what();
        return (int)r.error();
    }
    const size_t* l = r.value()->GetLims();
    for (int64_t i = 0; i <= nq; i++) lims[i] = (int64_t)l[i];
    const size_t n = l[nq];
    *ids = static_cast<int64_t*>(std::malloc(sizeof(int64_t) * (n + 1)));
    *dist = static_cast<float*>(std::malloc(sizeof(float) * (n + 1)));
    std::memcpy(*ids, r.value()->GetIds(), sizeof(int64_t) * n);
    std::memcpy(*dist, r.value()->GetDistance(), sizeof(float) * n);
    return 0;
}

// the node's NormalizeVec restatement on n rows in place (norms may be null): test hook for the bitwise check against
// knowhere::NormalizeVecs (src/common/utils.cc:60-93)
void knhip_node_normalize_rows(float* x, int64_t n, int64_t d, float* norms) { knhip_host_normalize_rows(x, n, d, norms); }

// Index::Train / Index::Add separately (Build = both): repeated Add is how a growing segment is fed
int knhip_node_train(void* h, const float* x, int64_t rows, int64_t dim, const char* cfg) {
    auto ds = GenDataSet(rows, dim, x);
    return (int)static_cast<Handle*>(h)->idx.Train(ds, ParseConfig(cfg));
}
int knhip_node_add(void* h, const float* x, int64_t rows, int64_t dim, const char* cfg) {
    auto ds = GenDataSet(rows, dim, x);
    return (int)static_cast<Handle*>(h)->idx.Add(ds, ParseConfig(cfg));
}

// Index::GetVectorByIds: out receives n * dim floats
int knhip_node_get_vectors(void* h, const int64_t* ids, int64_t n, int64_t dim, float* out) {
    auto ds = GenIdsDataSet(n, ids);
    auto r = static_cast<Handle*>(h)->idx.GetVectorByIds(ds);
    if (!r.has_value()) {
        g_err = r.what();
        return (int)r.error();
    }
    std::memcpy(out, r.value()->GetTensor(), sizeof(float) * (size_t)(n * dim));
    return 0;
}

// where the index built / loaded last on the calling thread was placed (device ordinals in shard order): test hook
int32_t knhip_node_last_placement(int32_t* out, int32_t cap) { return knhip_host_last_placement(out, cap); }

int64_t knhip_node_count(void* h) { return static_cast<Handle*>(h)->idx.Count(); }
int64_t knhip_node_dim(void* h) { return static_cast<Handle*>(h)->idx.Dim(); }

}  // extern "C"
