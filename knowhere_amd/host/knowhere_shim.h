// knowhere_amd/host/knowhere_shim.h -- stand-in for the Knowhere headers the host node compiles
// against.
//
// Real Knowhere cannot be built in this image (Conan dependencies milvus-common, folly, glog,
// nlohmann_json, ... are absent), so the IndexNode in hip_index_node.cc is compiled
// and tested against this header, which reproduces -- same names, same argument meaning, same
// error behaviour -- exactly the slice of the reference interface the node touches:
//   Status / expected<T>             include/knowhere/expected.h:34-68
//   DataSet, GenDataSet, GenResultDataSet   include/knowhere/dataset.h:412-512
//   BitsetView                        include/knowhere/bitsetview.h
//   Json + meta:: / indexparam:: / metric:: keys   include/knowhere/comp/index_param.h:42-164
//   IndexNode (pure virtuals)         include/knowhere/index/index_node.h:131-395
//   Index<T> facade                   include/knowhere/index/index.h:160-235, src/index/index.cc
//   IndexFactory + registration       include/knowhere/index/index_factory.h:29-165
//   BinarySet                         include/knowhere/binaryset.h
// INTEGRATION.md shows the two-line change that swaps this header for the real ones.
#pragma once

#include <atomic>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <variant>
#include <vector>

namespace knowhere {

// ---- expected.h ----------------------------------------------------------------------------------
enum class Status {
    success = 0,
    invalid_args = 1,
    invalid_param_in_json = 2,
    out_of_range_in_json = 3,
    type_conflict_in_json = 4,
    invalid_metric_type = 5,
    empty_index = 6,
    not_implemented = 7,
    index_not_trained = 8,
    index_already_trained = 9,
    faiss_inner_error = 10,
    malloc_error = 13,
    invalid_value_in_json = 16,
    invalid_binary_set = 19,
    cuda_runtime_error = 22,
    invalid_index_error = 23,
    internal_error = 27,
    invalid_serialized_index_type = 28,
    knowhere_inner_error = 33,
};

inline std::string Status2String(Status s) {
    switch (s) {
        case Status::success: return "success";
        case Status::invalid_args: return "invalid args";
        case Status::faiss_inner_error: return "faiss inner error";
        case Status::invalid_param_in_json: return "invalid param in json";
        case Status::out_of_range_in_json: return "out of range in json";
        case Status::type_conflict_in_json: return "type conflict in json";
        case Status::invalid_metric_type: return "invalid metric type";
        case Status::empty_index: return "empty index";
        case Status::not_implemented: return "not implemented";
        case Status::index_not_trained: return "index not trained";
        case Status::index_already_trained: return "index already trained";
        case Status::malloc_error: return "malloc error";
        case Status::invalid_value_in_json: return "invalid value in json";
        case Status::invalid_binary_set: return "invalid binary set";
        case Status::cuda_runtime_error: return "cuda runtime error";
        case Status::invalid_index_error: return "invalid index error";
        case Status::invalid_serialized_index_type: return "invalid serialized index type";
        default: return "internal error";
    }
}

template <typename T>
class expected {
 public:
    expected(const T& v) : val_(v), status_(Status::success) {}
    expected(T&& v) : val_(std::move(v)), status_(Status::success) {}
    expected(Status s) : status_(s) {}
    static expected<T> Err(Status s, std::string msg) {
        expected<T> e(s);
        e.msg_ = std::move(msg);
        return e;
    }
    bool has_value() const { return status_ == Status::success; }
    Status error() const { return status_; }
    const T& value() const {
        if (!has_value()) throw std::runtime_error("expected<T>::value() on error: " + msg_);
        return val_;
    }
    T& value() {
        if (!has_value()) throw std::runtime_error("expected<T>::value() on error: " + msg_);
        return val_;
    }
    const std::string& what() const { return msg_; }

 private:
    T val_{};
    Status status_;
    std::string msg_;
};

// ---- comp/index_param.h ---------------------------------------------------------------------------
namespace meta {
constexpr const char* METRIC_TYPE = "metric_type";
constexpr const char* DIM = "dim";
constexpr const char* ROWS = "rows";
constexpr const char* TOPK = "k";
constexpr const char* RADIUS = "radius";
constexpr const char* RANGE_FILTER = "range_filter";
}  // namespace meta
namespace indexparam {
constexpr const char* NPROBE = "nprobe";
constexpr const char* MAX_EMPTY_RESULT_BUCKETS = "max_empty_result_buckets";
constexpr const char* NLIST = "nlist";
constexpr const char* NBITS = "nbits";
constexpr const char* M = "m";
constexpr const char* REFINE = "refine";
constexpr const char* REFINE_K = "refine_k";
}  // namespace indexparam
namespace metric {
constexpr const char* L2 = "L2";
constexpr const char* IP = "IP";
constexpr const char* COSINE = "COSINE";
}  // namespace metric
namespace IndexEnum {
// new index types this backend registers (next to include/knowhere/comp/index_param.h:42-55)
constexpr const char* INDEX_HIP_BRUTEFORCE = "GPU_HIP_BRUTE_FORCE";
constexpr const char* INDEX_HIP_IVFFLAT = "GPU_HIP_IVF_FLAT";
constexpr const char* INDEX_HIP_IVFPQ = "GPU_HIP_IVF_PQ";
constexpr const char* INDEX_HIP_IVFSQ8 = "GPU_HIP_IVF_SQ8";
}  // namespace IndexEnum

struct fp32 {};  // data-type tag (include/knowhere/operands.h)

namespace Version {
inline int32_t GetCurrentVersion() { return 9; }
}  // namespace Version

// ---- Json (the slice of nlohmann::json the configs use) ---------------------------------------------
class JsonValue {
 public:
    using V = std::variant<std::monostate, bool, int64_t, double, std::string>;
    JsonValue() = default;
    JsonValue& operator=(bool v) { v_ = v; return *this; }
    JsonValue& operator=(int v) { v_ = (int64_t)v; return *this; }
    JsonValue& operator=(int64_t v) { v_ = v; return *this; }
    JsonValue& operator=(double v) { v_ = v; return *this; }
    JsonValue& operator=(float v) { v_ = (double)v; return *this; }
    JsonValue& operator=(const char* v) { v_ = std::string(v); return *this; }
    JsonValue& operator=(const std::string& v) { v_ = v; return *this; }
    bool is_null() const { return std::holds_alternative<std::monostate>(v_); }
    bool is_string() const { return std::holds_alternative<std::string>(v_); }
    bool is_number() const { return std::holds_alternative<int64_t>(v_) || std::holds_alternative<double>(v_); }
    bool is_integer() const { return std::holds_alternative<int64_t>(v_); }
    bool is_boolean() const { return std::holds_alternative<bool>(v_); }
    int64_t as_int() const { return is_integer() ? std::get<int64_t>(v_) : (int64_t)std::get<double>(v_); }
    double as_double() const { return is_integer() ? (double)std::get<int64_t>(v_) : std::get<double>(v_); }
    bool as_bool() const { return std::get<bool>(v_); }
    const std::string& as_string() const { return std::get<std::string>(v_); }

 private:
    V v_;
};

class Json {
 public:
    JsonValue& operator[](const std::string& k) { return m_[k]; }
    bool contains(const std::string& k) const { return m_.count(k) != 0; }
    const JsonValue& at(const std::string& k) const { return m_.at(k); }

 private:
    std::map<std::string, JsonValue> m_;
};

// ---- bitsetview.h -------------------------------------------------------------------------------------
class BitsetView {
 public:
    BitsetView() = default;
    BitsetView(std::nullptr_t) {}
    BitsetView(const uint8_t* data, size_t num_bits, size_t filtered_out = (size_t)-1)
        : bits_(data), num_bits_(num_bits), filtered_(filtered_out) {}
    bool empty() const { return num_bits_ == 0; }
    size_t size() const { return num_bits_; }
    size_t byte_size() const { return (num_bits_ + 7) >> 3; }
    const uint8_t* data() const { return bits_; }
    bool test(int64_t index) const { return bits_[index >> 3] & (0x1 << (index & 0x7)); }
    size_t count() const {
        if (filtered_ != (size_t)-1) return filtered_;
        size_t c = 0;
        for (size_t i = 0; i < num_bits_; i++) c += test((int64_t)i);
        return c;
    }

 private:
    const uint8_t* bits_ = nullptr;
    size_t num_bits_ = 0;
    size_t filtered_ = (size_t)-1;
};

// ---- dataset.h ------------------------------------------------------------------------------------------
class DataSet {
 public:
    ~DataSet() {
        if (owner_) {
            delete[] static_cast<const char*>(tensor_);
            delete[] ids_;
            delete[] dist_;
            delete[] lims_;
        }
    }
    void SetRows(int64_t r) { rows_ = r; }
    void SetDim(int64_t d) { dim_ = d; }
    void SetTensor(const void* t) { tensor_ = t; }
    void SetIds(const int64_t* i) { ids_ = i; }
    void SetDistance(const float* d) { dist_ = d; }
    void SetIsOwner(bool o) { owner_ = o; }
    void SetLims(const size_t* l) { lims_ = l; }
    const size_t* GetLims() const { return lims_; }
    void SetTensorBeginId(int64_t b) { begin_id_ = b; }
    int64_t GetRows() const { return rows_; }
    int64_t GetDim() const { return dim_; }
    const void* GetTensor() const { return tensor_; }
    const int64_t* GetIds() const { return ids_; }
    const float* GetDistance() const { return dist_; }
    int64_t GetTensorBeginId() const { return begin_id_; }

 private:
    int64_t rows_ = 0, dim_ = 0, begin_id_ = 0;
    const void* tensor_ = nullptr;
    const int64_t* ids_ = nullptr;
    const float* dist_ = nullptr;
    const size_t* lims_ = nullptr;
    bool owner_ = true;
};
using DataSetPtr = std::shared_ptr<DataSet>;

inline DataSetPtr GenDataSet(int64_t rows, int64_t dim, const void* tensor) {
    auto ds = std::make_shared<DataSet>();
    ds->SetRows(rows);
    ds->SetDim(dim);
    ds->SetTensor(tensor);
    ds->SetIsOwner(false);
    return ds;
}

/// takes ownership of two new[]-allocated arrays (include/knowhere/dataset.h:497-512)
inline DataSetPtr GenResultDataSet(int64_t nq, int64_t topk, const int64_t* ids, const float* distance) {
    auto ds = std::make_shared<DataSet>();
    ds->SetRows(nq);
    ds->SetDim(topk);
    ds->SetIds(ids);
    ds->SetDistance(distance);
    ds->SetIsOwner(true);
    return ds;
}

/// range search result (include/knowhere/dataset.h GenResultDataSet(nq, RangeSearchResult)): lims[nq + 1] +
/// flat ids / distances; takes ownership of three new[]-allocated arrays
inline DataSetPtr GenRangeResultDataSet(int64_t nq, const size_t* lims, const int64_t* ids, const float* distance) {
    auto ds = std::make_shared<DataSet>();
    ds->SetRows(nq);
    ds->SetLims(lims);
    ds->SetIds(ids);
    ds->SetDistance(distance);
    ds->SetIsOwner(true);
    return ds;
}

// ---- binaryset.h ----------------------------------------------------------------------------------------
struct Binary {
    std::shared_ptr<uint8_t[]> data;
    int64_t size = 0;
};
using BinaryPtr = std::shared_ptr<Binary>;
class BinarySet {
 public:
    BinaryPtr GetByName(const std::string& name) const {
        auto it = m_.find(name);
        return it == m_.end() ? nullptr : it->second;
    }
    void Append(const std::string& name, std::shared_ptr<uint8_t[]> data, int64_t size) {
        auto b = std::make_shared<Binary>();
        b->data = std::move(data);
        b->size = size;
        m_[name] = b;
    }
    bool Contains(const std::string& name) const { return m_.count(name) != 0; }

 private:
    std::map<std::string, BinaryPtr> m_;
};

// ---- index_node.h -----------------------------------------------------------------------------------------
class IndexNode {
 public:
    virtual ~IndexNode() = default;
    virtual Status Build(const DataSetPtr dataset, const Json& cfg) {
        Status s = Train(dataset, cfg);
        if (s != Status::success) return s;
        return Add(dataset, cfg);
    }
    virtual Status Train(const DataSetPtr dataset, const Json& cfg) = 0;
    virtual Status Add(const DataSetPtr dataset, const Json& cfg) = 0;
    virtual expected<DataSetPtr> Search(const DataSetPtr dataset, const Json& cfg, const BitsetView& bitset) const = 0;
    virtual expected<DataSetPtr> RangeSearch(const DataSetPtr dataset, const Json& cfg,
                                             const BitsetView& bitset) const = 0;
    virtual expected<DataSetPtr> GetVectorByIds(const DataSetPtr dataset) const = 0;
    virtual bool HasRawData(const std::string& metric_type) const = 0;
    virtual expected<DataSetPtr> GetIndexMeta(const Json& cfg) const = 0;
    virtual Status Serialize(BinarySet& binset) const = 0;
    virtual Status Deserialize(const BinarySet& binset, const Json& cfg) = 0;
    virtual Status DeserializeFromFile(const std::string& filename, const Json& cfg) = 0;
    virtual int64_t Dim() const = 0;
    virtual int64_t Size() const = 0;
    virtual int64_t Count() const = 0;
    virtual std::string Type() const = 0;
};

// ---- index.h: ref-counted facade; every call is guarded (exceptions -> Status), as GuardedCall does ---------
template <typename T1>
class Index {
 public:
    Index() = default;
    explicit Index(std::shared_ptr<T1> node) : node_(std::move(node)) {}
    template <class F>
    static Status Guard(F&& f) {
        try {
            return f();
        } catch (const std::bad_alloc&) {
            return Status::malloc_error;
        } catch (...) {
            return Status::knowhere_inner_error;
        }
    }
    Status Build(const DataSetPtr ds, const Json& cfg) {
        return Guard([&] { return node_->Build(ds, cfg); });
    }
    Status Train(const DataSetPtr ds, const Json& cfg) {
        return Guard([&] { return node_->Train(ds, cfg); });
    }
    Status Add(const DataSetPtr ds, const Json& cfg) {
        return Guard([&] { return node_->Add(ds, cfg); });
    }
    expected<DataSetPtr> Search(const DataSetPtr ds, const Json& cfg, const BitsetView& bitset) const {
        try {
            return node_->Search(ds, cfg, bitset);
        } catch (const std::exception& e) {
            return expected<DataSetPtr>::Err(Status::knowhere_inner_error, e.what());
        }
    }
    expected<DataSetPtr> RangeSearch(const DataSetPtr ds, const Json& cfg, const BitsetView& bitset) const {
        return node_->RangeSearch(ds, cfg, bitset);
    }
    expected<DataSetPtr> GetVectorByIds(const DataSetPtr ds) const { return node_->GetVectorByIds(ds); }
    bool HasRawData(const std::string& m) const { return node_->HasRawData(m); }
    Status Serialize(BinarySet& b) const {
        return Guard([&] { return node_->Serialize(b); });
    }
    Status Deserialize(const BinarySet& b, const Json& cfg = Json()) {
        return Guard([&] { return node_->Deserialize(b, cfg); });
    }
    int64_t Dim() const { return node_->Dim(); }
    int64_t Size() const { return node_->Size(); }
    int64_t Count() const { return node_->Count(); }
    std::string Type() const { return node_->Type(); }
    T1* Node() const { return node_.get(); }

 private:
    std::shared_ptr<T1> node_;
};

// ---- index_factory.h ----------------------------------------------------------------------------------------
class IndexFactory {
 public:
    using Creator = std::function<Index<IndexNode>(const int32_t&)>;
    static IndexFactory& Instance() {
        static IndexFactory f;
        return f;
    }
    template <typename DataType>
    expected<Index<IndexNode>> Create(const std::string& name, const int32_t& version) {
        std::lock_guard<std::mutex> lk(mu_);
        auto it = map_.find(name);
        if (it == map_.end()) {
            return expected<Index<IndexNode>>::Err(Status::invalid_index_error, "failed to find index " + name);
        }
        return it->second(version);
    }
    const IndexFactory& Register(const std::string& name, Creator c) {
        std::lock_guard<std::mutex> lk(mu_);
        map_[name] = std::move(c);
        return *this;
    }

 private:
    std::mutex mu_;
    std::map<std::string, Creator> map_;
};

// cf. KNOWHERE_REGISTER_GLOBAL_WITH_THREAD_POOL (include/knowhere/index/index_factory.h:157-165); the
// real macro also wraps the node in IndexNodeThreadPoolWrapper to bound in-flight GPU searches
#define KNOWHERE_HIP_REGISTER_GLOBAL(name, NodeType, ...)                                                     \
    static const ::knowhere::IndexFactory& name##_reg_ref = ::knowhere::IndexFactory::Instance().Register(    \
            #name, [](const int32_t& version) {                                                                \
                return ::knowhere::Index<::knowhere::IndexNode>(std::make_shared<NodeType>(version, ##__VA_ARGS__)); \
            })

// ---- comp/brute_force.h ---------------------------------------------------------------------------------------
struct BruteForce {
    template <typename DataType>
    static expected<DataSetPtr> Search(const DataSetPtr base_dataset, const DataSetPtr query_dataset,
                                       const Json& config, const BitsetView& bitset);
};

}  // namespace knowhere
