// knowhere_amd/host/knowhere_shim.h -- stand-in for the Knowhere headers the host node compiles against WHEN THE
// REFERENCE TREE IS NOT THERE (the GPU box; the reference's Conan dependencies -- milvus-common, folly, glog,
// nlohmann_json 3.11, boost -- are absent from this image).
//
// hip_index_node.{h,cc} are written against the REAL Knowhere interface and are compile-checked against the
// reference's own headers (tests/test_node_contract.py: `-I/root/reference/include`, stubs only for the absent
// third-party headers).  This header reproduces, name for name and signature for signature, the slice of that
// interface the node and its tests touch, so the same sources also build and RUN here:
//   Status / expected<T> / RETURN_IF_ERROR        include/knowhere/expected.h:34-68, 262-400
//   Config machinery (CFG_*, Entry, EntryAccess,  include/knowhere/config.h:38-600
//     KNOWHERE_DECLARE_CONFIG, KNOWHERE_CONFIG_DECLARE_FIELD, Config::Load, BaseConfig)
//   IvfConfig / IvfFlatConfig / IvfPqConfig /     src/index/ivf/ivf_config.h:24-260, src/index/flat/flat_config.h
//     IvfSqConfig / FlatConfig
//   DataSet + Gen*DataSet                          include/knowhere/dataset.h:412-560
//   BitsetView (incl. the out-id view)             include/knowhere/bitsetview.h:38-180
//   BinarySet, Object, Version, feature flags      include/knowhere/{binaryset,object,version,feature}.h
//   IndexNode (virtuals as in the reference)       include/knowhere/index/index_node.h:100-400, 770-800
//   IndexNodeThreadPoolWrapper                     include/knowhere/index/index_node_thread_pool_wrapper.h
//   Index<T> facade                                include/knowhere/index/index.h:160-235, src/index/index.cc
//   IndexFactory / IndexStaticFaced + the          include/knowhere/index/index_factory.h:29-165,
//     KNOWHERE_REGISTER_* macros (same text)       include/knowhere/index/index_static.h:47-140
//   milvus::OpContext / checkCancellation          include/knowhere/context.h:21-33
#pragma once

#include <atomic>
#include <cassert>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <variant>
#include <vector>

// ---- common/OpContext.h (milvus-common) + context.h ---------------------------------------------------------------
namespace milvus {
struct OpContext {
    std::atomic<bool> cancelled{false};  // (the reference holds a folly::CancellationToken)
};
}  // namespace milvus

namespace knowhere {

struct OperationCancelled : std::runtime_error {  // (the reference throws folly::FutureCancellation)
    OperationCancelled() : std::runtime_error("operation cancelled") {}
};
inline void
checkCancellation(const milvus::OpContext* op_context) {
    if (op_context != nullptr && op_context->cancelled.load()) {
        throw OperationCancelled();
    }
}

// ---- expected.h ---------------------------------------------------------------------------------------------------
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
    hnsw_inner_error = 12,
    malloc_error = 13,
    diskann_inner_error = 14,
    disk_file_error = 15,
    invalid_value_in_json = 16,
    arithmetic_overflow = 17,
    cuvs_inner_error = 18,
    invalid_binary_set = 19,
    invalid_instruction_set = 20,
    cardinal_inner_error = 21,
    cuda_runtime_error = 22,
    invalid_index_error = 23,
    invalid_cluster_error = 24,
    cluster_inner_error = 25,
    timeout = 26,
    internal_error = 27,
    invalid_serialized_index_type = 28,
    sparse_inner_error = 29,
    brute_force_inner_error = 30,
    emb_list_inner_error = 31,
    minhash_inner_error = 32,
    knowhere_inner_error = 33,
};

inline std::string
Status2String(Status s) {
    switch (s) {
        case Status::success: return "success";
        case Status::invalid_args: return "invalid args";
        case Status::invalid_param_in_json: return "invalid param in json";
        case Status::out_of_range_in_json: return "out of range in json";
        case Status::type_conflict_in_json: return "type conflict in json";
        case Status::invalid_metric_type: return "invalid metric type";
        case Status::empty_index: return "empty index";
        case Status::not_implemented: return "not implemented";
        case Status::index_not_trained: return "index not trained";
        case Status::index_already_trained: return "index already trained";
        case Status::faiss_inner_error: return "faiss inner error";
        case Status::malloc_error: return "malloc error";
        case Status::invalid_value_in_json: return "invalid value in json";
        case Status::arithmetic_overflow: return "arithmetic overflow";
        case Status::invalid_binary_set: return "invalid binary set";
        case Status::cuda_runtime_error: return "cuda runtime error";
        case Status::invalid_index_error: return "invalid index error";
        case Status::invalid_serialized_index_type: return "invalid serialized index type";
        case Status::timeout: return "timeout";
        default: return "internal error";
    }
}

template <typename T>
class expected {
 public:
    expected(const T& v) : val_(v), status_(Status::success) {}
    expected(T&& v) : val_(std::move(v)), status_(Status::success) {}
    expected(Status s) : status_(s) { assert(s != Status::success); }
    static expected<T>
    Err(Status s, std::string msg) {
        expected<T> e(s);
        e.msg_ = std::move(msg);
        return e;
    }
    bool has_value() const { return status_ == Status::success; }
    Status error() const { return status_; }
    const T& value() const {
        if (!has_value()) throw std::runtime_error("expected<T>::value() on error: " + msg_);
        return *val_;
    }
    T& value() {
        if (!has_value()) throw std::runtime_error("expected<T>::value() on error: " + msg_);
        return *val_;
    }
    const std::string& what() const { return msg_; }

 private:
    std::optional<T> val_;
    Status status_;
    std::string msg_;
};

#define RETURN_IF_ERROR(expr)                       \
    do {                                            \
        auto status_tmp_ = (expr);                  \
        if (status_tmp_ != ::knowhere::Status::success) { \
            return status_tmp_;                     \
        }                                           \
    } while (0)

// ---- log.h ----------------------------------------------------------------------------------------------------------
struct NullLog {
    template <class T>
    NullLog& operator<<(const T&) { return *this; }
};
#define LOG_KNOWHERE_ERROR_ ::knowhere::NullLog()
#define LOG_KNOWHERE_WARNING_ ::knowhere::NullLog()
#define LOG_KNOWHERE_INFO_ ::knowhere::NullLog()

// ---- comp/index_param.h ---------------------------------------------------------------------------------------------
using IndexType = std::string;
using MetricType = std::string;
using IndexVersion = int32_t;
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
constexpr const char* REFINE_TYPE = "refine_type";
constexpr const char* SQ_TYPE = "sq_type";  // for IVF_SQ and HNSW_SQ (index_param.h:238)
constexpr const char* IVF_SQ_TYPE = "sq_type";
}  // namespace indexparam
namespace metric {
constexpr const char* L2 = "L2";
constexpr const char* IP = "IP";
constexpr const char* COSINE = "COSINE";
}  // namespace metric
inline bool
IsMetricType(const std::string& str, const knowhere::MetricType& metric_type) {
    if (str.size() != metric_type.size()) return false;
    for (size_t i = 0; i < str.size(); i++) {
        if (std::toupper((unsigned char)str[i]) != std::toupper((unsigned char)metric_type[i])) return false;
    }
    return true;
}

// ---- operands.h / feature.h / object.h / version.h --------------------------------------------------------------------
struct fp32 {};
struct fp16 {};
struct bf16 {};
struct int8 {};
struct bin1 {};

namespace feature {
constexpr uint64_t BINARY = 1UL << 0;
constexpr uint64_t FLOAT32 = 1UL << 1;
constexpr uint64_t FP16 = 1UL << 2;
constexpr uint64_t BF16 = 1UL << 3;
constexpr uint64_t SPARSE_U32_F32 = 1UL << 4;
constexpr uint64_t INT8 = 1UL << 5;
constexpr uint64_t NO_TRAIN = 1UL << 16;
constexpr uint64_t KNN = 1UL << 17;
constexpr uint64_t GPU = 1UL << 18;
constexpr uint64_t MMAP = 1UL << 19;
constexpr uint64_t GPU_KNN_FLOAT_INDEX = FLOAT32 | GPU | KNN;
constexpr uint64_t GPU_ANN_FLOAT_INDEX = FLOAT32 | GPU;
}  // namespace feature

template <typename DataType>
inline bool
typeCheck(uint64_t features) {
    if constexpr (std::is_same_v<DataType, fp32>) return features & feature::FLOAT32;
    if constexpr (std::is_same_v<DataType, fp16>) return features & feature::FP16;
    if constexpr (std::is_same_v<DataType, bf16>) return features & feature::BF16;
    if constexpr (std::is_same_v<DataType, int8>) return features & feature::INT8;
    if constexpr (std::is_same_v<DataType, bin1>) return features & feature::BINARY;
    return false;
}

class Object {
 public:
    Object() = default;
    Object(const std::nullptr_t) {}
    virtual ~Object() {}
};

class Version {
 public:
    explicit Version(IndexVersion v) : version_(v) {}
    static Version GetCurrentVersion() { return Version(9); }
    IndexVersion VersionNumber() const { return version_; }

 private:
    IndexVersion version_;
};

// ---- Json: the slice of nlohmann::json the configs and tests use ------------------------------------------------------
class JsonValue {
 public:
    using V = std::variant<std::monostate, bool, int64_t, double, std::string>;
    JsonValue() = default;
    JsonValue& operator=(bool v) { v_ = v; return *this; }
    JsonValue& operator=(int v) { v_ = (int64_t)v; return *this; }
    JsonValue& operator=(int64_t v) { v_ = v; return *this; }
    JsonValue& operator=(size_t v) { v_ = (int64_t)v; return *this; }
    JsonValue& operator=(double v) { v_ = v; return *this; }
    JsonValue& operator=(float v) { v_ = (double)v; return *this; }
    JsonValue& operator=(const char* v) { v_ = std::string(v); return *this; }
    JsonValue& operator=(const std::string& v) { v_ = v; return *this; }
    bool is_null() const { return std::holds_alternative<std::monostate>(v_); }
    bool is_string() const { return std::holds_alternative<std::string>(v_); }
    bool is_number() const { return std::holds_alternative<int64_t>(v_) || std::holds_alternative<double>(v_); }
    bool is_number_integer() const { return std::holds_alternative<int64_t>(v_); }
    bool is_boolean() const { return std::holds_alternative<bool>(v_); }
    int64_t as_int() const { return is_number_integer() ? std::get<int64_t>(v_) : (int64_t)std::get<double>(v_); }
    double as_double() const { return is_number_integer() ? (double)std::get<int64_t>(v_) : std::get<double>(v_); }
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
    void erase(const std::string& k) { m_.erase(k); }

 private:
    std::map<std::string, JsonValue> m_;
};

// ---- config.h -----------------------------------------------------------------------------------------------------------
#define CFG_INT std::optional<int32_t>
#define CFG_INT64 std::optional<int64_t>
#define CFG_STRING std::optional<std::string>
#define CFG_FLOAT std::optional<float>
#define CFG_BOOL std::optional<bool>

template <typename T>
struct Range {
    Range(T l, T r, bool il, bool ir) : left(l), right(r), include_left(il), include_right(ir) {}
    bool within(T v) const {
        return (include_left ? v >= left : v > left) && (include_right ? v <= right : v < right);
    }
    T left, right;
    bool include_left, include_right;
};

enum PARAM_TYPE {
    TRAIN = 1 << 0,
    SEARCH = 1 << 1,
    RANGE_SEARCH = 1 << 2,
    FEDER = 1 << 3,
    DESERIALIZE = 1 << 4,
    DESERIALIZE_FROM_FILE = 1 << 5,
    ITERATOR = 1 << 6,
    CLUSTER = 1 << 7,
    STATIC = 1 << 8,
};

template <typename T>
struct Entry {
    explicit Entry(T* v) : val(v) {}
    Entry() = default;
    T* val = nullptr;
    uint32_t type = 0;
    std::optional<typename T::value_type> default_val;
    std::optional<Range<typename T::value_type>> range;  // (numeric entries)
    std::optional<std::string> desc;
    bool allow_empty_without_default = false;
};

template <typename T>
class EntryAccess {
 public:
    EntryAccess(Entry<T>* entry) : entry(entry) {}
    EntryAccess& set_default(const typename T::value_type dft) {
        entry->default_val = dft;
        *entry->val = dft;
        return *this;
    }
    EntryAccess& set_range(typename T::value_type a, typename T::value_type b, bool include_left = true,
                           bool include_right = true) {
        entry->range = Range<typename T::value_type>(a, b, include_left, include_right);
        return *this;
    }
    EntryAccess& allow_empty_without_default() { entry->allow_empty_without_default = true; return *this; }
    EntryAccess& description(const std::string& desc) { entry->desc = desc; return *this; }
    EntryAccess& for_static() { entry->type |= PARAM_TYPE::STATIC; return *this; }
    EntryAccess& for_train() { entry->type |= PARAM_TYPE::TRAIN; return *this; }
    EntryAccess& for_search() { entry->type |= PARAM_TYPE::SEARCH; return *this; }
    EntryAccess& for_range_search() { entry->type |= PARAM_TYPE::RANGE_SEARCH; return *this; }
    EntryAccess& for_iterator() { entry->type |= PARAM_TYPE::ITERATOR; return *this; }
    EntryAccess& for_feder() { entry->type |= PARAM_TYPE::FEDER; return *this; }
    EntryAccess& for_cluster() { entry->type |= PARAM_TYPE::CLUSTER; return *this; }
    EntryAccess& for_deserialize() { entry->type |= PARAM_TYPE::DESERIALIZE; return *this; }
    EntryAccess& for_deserialize_from_file() { entry->type |= PARAM_TYPE::DESERIALIZE_FROM_FILE; return *this; }
    EntryAccess& for_train_and_search() {
        entry->type |= PARAM_TYPE::TRAIN | PARAM_TYPE::SEARCH | PARAM_TYPE::RANGE_SEARCH;
        return *this;
    }

 private:
    Entry<T>* entry;
};

class Config {
 public:
    // string values of numeric / boolean parameters are converted in place (Milvus passes strings), as the
    // reference's FormatAndCheck does (src/common/config.cc)
    static Status
    FormatAndCheck(const Config& cfg, Json& json, std::string* const err_msg = nullptr) {
        for (const auto& it : cfg.__DICT__) {
            if (!json.contains(it.first) || !json.at(it.first).is_string()) continue;
            const std::string sv = json.at(it.first).as_string();
            try {
                if (std::get_if<Entry<CFG_INT>>(&it.second) || std::get_if<Entry<CFG_INT64>>(&it.second)) {
                    size_t pos = 0;
                    const int64_t v = std::stoll(sv, &pos);
                    if (pos != sv.size()) throw std::invalid_argument("trailing characters");
                    json[it.first] = v;
                } else if (std::get_if<Entry<CFG_FLOAT>>(&it.second)) {
                    json[it.first] = std::stod(sv);
                } else if (std::get_if<Entry<CFG_BOOL>>(&it.second)) {
                    if (sv == "true" || sv == "True") json[it.first] = true;
                    else if (sv == "false" || sv == "False") json[it.first] = false;
                    else throw std::invalid_argument("not a boolean");
                }
            } catch (const std::exception&) {
                if (err_msg) *err_msg = "invalid value in json for param " + it.first;
                return Status::invalid_value_in_json;
            }
        }
        return Status::success;
    }

    static Status
    Load(Config& cfg, const Json& json, PARAM_TYPE type, std::string* const err_msg = nullptr) {
        auto fail = [&](Status s, const std::string& m) {
            if (err_msg) *err_msg = m;
            return s;
        };
        for (auto& it : cfg.__DICT__) {
            const std::string& name = it.first;
            Status st = Status::success;
            std::visit(
                [&](auto& e) {
                    using E = std::decay_t<decltype(e)>;
                    using T = std::remove_pointer_t<decltype(e.val)>;
                    using VT = typename T::value_type;
                    if (!(e.type & type)) return;
                    if (!json.contains(name) || json.at(name).is_null()) {
                        if (e.default_val.has_value()) {
                            *e.val = e.default_val;
                        } else if (e.allow_empty_without_default) {
                            *e.val = std::nullopt;
                        } else {
                            st = fail(Status::invalid_param_in_json, "param '" + name + "' not exist in json");
                        }
                        return;
                    }
                    const JsonValue& v = json.at(name);
                    if constexpr (std::is_same_v<VT, std::string>) {
                        if (!v.is_string()) {
                            st = fail(Status::type_conflict_in_json, "Type conflict in json: param '" + name +
                                                                         "' should be a string");
                            return;
                        }
                        *e.val = v.as_string();
                    } else if constexpr (std::is_same_v<VT, bool>) {
                        if (!v.is_boolean()) {
                            st = fail(Status::type_conflict_in_json, "Type conflict in json: param '" + name +
                                                                         "' should be a boolean");
                            return;
                        }
                        *e.val = v.as_bool();
                    } else if constexpr (std::is_same_v<VT, float>) {
                        if (!v.is_number()) {
                            st = fail(Status::type_conflict_in_json, "Type conflict in json: param '" + name +
                                                                         "' should be a number");
                            return;
                        }
                        const double d = v.as_double();
                        if (e.range.has_value() && !e.range->within((float)d)) {
                            st = fail(Status::out_of_range_in_json, "Out of range in json: param '" + name + "'");
                            return;
                        }
                        *e.val = (float)d;
                    } else {
                        if (!v.is_number_integer()) {
                            st = fail(Status::type_conflict_in_json, "Type conflict in json: param '" + name +
                                                                         "' should be integer");
                            return;
                        }
                        const int64_t iv = v.as_int();
                        if (iv > (int64_t)std::numeric_limits<VT>::max() || iv < (int64_t)std::numeric_limits<VT>::min()) {
                            st = fail(Status::arithmetic_overflow, "Arithmetic overflow: param '" + name + "'");
                            return;
                        }
                        if (e.range.has_value() && !e.range->within((VT)iv)) {
                            st = fail(Status::out_of_range_in_json, "Out of range in json: param '" + name + "'");
                            return;
                        }
                        *e.val = (VT)iv;
                    }
                    (void)sizeof(E);
                },
                it.second);
            if (st != Status::success) return st;
        }
        return cfg.CheckAndAdjust(type, err_msg);
    }

    virtual ~Config() {}

    using VarEntry =
        std::variant<Entry<CFG_STRING>, Entry<CFG_FLOAT>, Entry<CFG_INT>, Entry<CFG_INT64>, Entry<CFG_BOOL>>;
    std::unordered_map<std::string, VarEntry> __DICT__;

 protected:
    inline virtual Status
    CheckAndAdjust(PARAM_TYPE param_type, std::string* const err_msg) {
        return Status::success;
    }

    static knowhere::Status
    HandleError(std::string* error_msg, const std::string& msg, const knowhere::Status& status) {
        if (error_msg) *error_msg = msg;
        return status;
    }
};

#define KNOWHERE_DECLARE_CONFIG(CONFIG) CONFIG()

#define KNOWHERE_CONFIG_DECLARE_FIELD(PARAM)                                                                     \
    __DICT__[#PARAM] = knowhere::Config::VarEntry(std::in_place_type<knowhere::Entry<decltype(PARAM)>>, &PARAM); \
    knowhere::EntryAccess<decltype(PARAM)> PARAM##_access(                                                       \
        std::get_if<knowhere::Entry<decltype(PARAM)>>(&__DICT__[#PARAM]));                                       \
    PARAM##_access

const float defaultRangeFilter = std::numeric_limits<float>::infinity();

class BaseConfig : public Config {
 public:
    CFG_INT64 dim;  // just used for config verify
    CFG_STRING metric_type;
    CFG_INT k;
    CFG_INT num_build_thread;
    // for distance metrics, we search for vectors with distance in [range_filter, radius).
    // for similarity metrics, we search for vectors with similarity in (radius, range_filter].
    CFG_FLOAT radius;
    CFG_INT range_search_k;
    CFG_FLOAT range_filter;
    CFG_BOOL enable_mmap;
    KNOWHERE_DECLARE_CONFIG(BaseConfig) {
        KNOWHERE_CONFIG_DECLARE_FIELD(dim).allow_empty_without_default().description("vector dim").for_train();
        KNOWHERE_CONFIG_DECLARE_FIELD(metric_type)
            .set_default("L2")
            .description("metric type")
            .for_train_and_search()
            .for_iterator()
            .for_deserialize();
        KNOWHERE_CONFIG_DECLARE_FIELD(k)
            .set_default(10)
            .description("search for top k similar vector.")
            .set_range(1, std::numeric_limits<CFG_INT::value_type>::max())
            .for_search();
        KNOWHERE_CONFIG_DECLARE_FIELD(num_build_thread)
            .description("index thread limit for build.")
            .allow_empty_without_default()
            .for_train();
        KNOWHERE_CONFIG_DECLARE_FIELD(radius).set_default(0.0).description("radius for range search").for_range_search();
        KNOWHERE_CONFIG_DECLARE_FIELD(range_search_k).set_default(-1).description("range search k").for_range_search();
        KNOWHERE_CONFIG_DECLARE_FIELD(range_filter)
            .set_default(defaultRangeFilter)
            .description("result filter for range search")
            .for_range_search();
        KNOWHERE_CONFIG_DECLARE_FIELD(enable_mmap)
            .set_default(false)
            .description("enable mmap for load index")
            .for_static()
            .for_deserialize()
            .for_deserialize_from_file();
    }
};

// ---- src/index/flat/flat_config.h, src/index/ivf/ivf_config.h ---------------------------------------------------------
class FlatConfig : public BaseConfig {};

class IvfConfig : public BaseConfig {
 public:
    CFG_INT nlist;
    CFG_INT nprobe;
    CFG_BOOL use_elkan;
    CFG_BOOL ensure_topk_full;
    CFG_INT max_empty_result_buckets;
    KNOWHERE_DECLARE_CONFIG(IvfConfig) {
        KNOWHERE_CONFIG_DECLARE_FIELD(nlist)
            .description("number of inverted lists.")
            .set_default(128)
            .for_train()
            .set_range(1, 65536);
        KNOWHERE_CONFIG_DECLARE_FIELD(nprobe)
            .set_default(8)
            .description("number of probes at query time.")
            .for_search()
            .for_range_search()
            .for_iterator()
            .set_range(1, 65536);
        KNOWHERE_CONFIG_DECLARE_FIELD(use_elkan).set_default(true).description("whether to use elkan algorithm").for_train();
        KNOWHERE_CONFIG_DECLARE_FIELD(ensure_topk_full)
            .set_default(true)
            .description("whether to make sure topk results full")
            .for_search();
        KNOWHERE_CONFIG_DECLARE_FIELD(max_empty_result_buckets)
            .set_default(2)
            .description("the maximum of continuous buckets with empty result")
            .for_range_search()
            .set_range(0, 65536);
    }
};

class IvfFlatConfig : public IvfConfig {};

class IvfPqConfig : public IvfConfig {
 public:
    CFG_INT m;
    CFG_INT nbits;
    CFG_BOOL refine;
    CFG_FLOAT refine_k;
    CFG_STRING refine_type;
    KNOWHERE_DECLARE_CONFIG(IvfPqConfig) {
        KNOWHERE_CONFIG_DECLARE_FIELD(m).description("m").for_train().set_range(1, 65536);
        KNOWHERE_CONFIG_DECLARE_FIELD(nbits).description("nbits").set_default(8).for_train().set_range(1, 24);
        KNOWHERE_CONFIG_DECLARE_FIELD(refine)
            .description("whether the refine is used during the train")
            .set_default(false)
            .for_train()
            .for_static();
        KNOWHERE_CONFIG_DECLARE_FIELD(refine_k)
            .description("refine k")
            .set_default(1)
            .set_range(1, std::numeric_limits<CFG_FLOAT::value_type>::max())
            .for_search();
        KNOWHERE_CONFIG_DECLARE_FIELD(refine_type)
            .description("the type of a refine index")
            .allow_empty_without_default()
            .for_train()
            .for_static();
    }
    Status
    CheckAndAdjust(PARAM_TYPE param_type, std::string* err_msg) override {
        if (param_type == PARAM_TYPE::TRAIN && dim.has_value() && m.has_value() && m.value() > 0 &&
            dim.value() % m.value() != 0) {
            return HandleError(err_msg, "The dimension of a vector (dim) should be a multiple of the number of "
                                        "subquantizers (m).", Status::invalid_args);
        }
        return Status::success;
    }
};

class IvfSqConfig : public IvfConfig {
 public:
    CFG_STRING sq_type;
    CFG_BOOL refine;
    CFG_FLOAT refine_k;
    CFG_STRING refine_type;
    KNOWHERE_DECLARE_CONFIG(IvfSqConfig) {
        KNOWHERE_CONFIG_DECLARE_FIELD(sq_type).description("the type of sq").set_default("SQ8").for_train().for_static();
        KNOWHERE_CONFIG_DECLARE_FIELD(refine)
            .description("whether the refine is used during the train")
            .set_default(false)
            .for_train()
            .for_static();
        KNOWHERE_CONFIG_DECLARE_FIELD(refine_k)
            .description("refine k")
            .set_default(1)
            .set_range(1, std::numeric_limits<CFG_FLOAT::value_type>::max())
            .for_search();
        KNOWHERE_CONFIG_DECLARE_FIELD(refine_type)
            .description("the type of a refine index")
            .allow_empty_without_default()
            .for_train()
            .for_static();
    }
};

// ---- bitsetview.h ---------------------------------------------------------------------------------------------------------
// Non-owning filter view.  bits_ is addressed by PUBLIC ids; backend selectors pass internal ids to test(); an
// installed out-id view (internal id -> public id, PrepareBitset) translates them.
class BitsetView {
 public:
    BitsetView() = default;
    BitsetView(const uint8_t* data, size_t num_bits, std::optional<size_t> filtered_count = std::nullopt)
        : bits_(data), num_bits_(num_bits), vector_count_(num_bits), filtered_count_(filtered_count) {}
    BitsetView(const std::nullptr_t) : BitsetView() {}
    bool empty() const { return num_bits_ == 0 || (filtered_count_.has_value() && filtered_count_.value() == 0); }
    size_t size() const { return vector_count_; }
    bool has_known_count() const { return num_bits_ == 0 || filtered_count_.has_value(); }
    size_t count() const {
        if (num_bits_ == 0) return 0;
        if (!filtered_count_.has_value()) throw std::logic_error("BitsetView filtered count is unknown");
        return filtered_count_.value();
    }
    size_t byte_size() const { return (num_bits_ + 8 - 1) >> 3; }
    size_t num_bits() const { return num_bits_; }
    const uint8_t* data() const { return bits_; }
    bool has_out_ids() const { return out_ids_count_ != 0; }
    size_t out_ids_count() const { return out_ids_count_; }
    void set_out_ids(const int64_t* out_ids, size_t out_ids_count) {
        out_ids_ = out_ids;
        out_ids_count_ = out_ids_count;
        vector_count_ = out_ids_count;
    }
    // recomputes the filtered count in the backend id domain (NeedBitsetExactCount backends)
    void count_filtered_bits() {
        size_t c = 0;
        for (size_t i = 0; i < vector_count_; i++) c += test((int64_t)i);
        filtered_count_ = c;
    }
    bool test(int64_t index) const {
        int64_t out_id = index;
        if (has_out_ids()) {
            if ((size_t)index >= out_ids_count_) return true;
            out_id = out_ids_[index];
        }
        if (out_id < 0 || (size_t)out_id >= num_bits_) return out_id < 0;
        return bits_[out_id >> 3] & (0x1 << (out_id & 0x7));
    }

 private:
    const uint8_t* bits_ = nullptr;
    size_t num_bits_ = 0;
    size_t vector_count_ = 0;
    std::optional<size_t> filtered_count_;
    const int64_t* out_ids_ = nullptr;
    size_t out_ids_count_ = 0;
};

// ---- id_map.h: backend storage id <-> public id (identity unless a map is installed) -----------------------------------
class IdMap {
 public:
    void SetInToOut(std::vector<int64_t> in_to_out) { in_to_out_ = std::move(in_to_out); }
    int64_t OutCount() const { return (int64_t)in_to_out_.size(); }
    bool Empty() const { return in_to_out_.empty(); }
    const int64_t* InToOut() const { return in_to_out_.data(); }
    void MapInToOut(int64_t* ids, size_t count) const {
        if (in_to_out_.empty()) return;
        for (size_t i = 0; i < count; i++) {
            if (ids[i] >= 0 && (size_t)ids[i] < in_to_out_.size()) ids[i] = in_to_out_[ids[i]];
        }
    }

 private:
    std::vector<int64_t> in_to_out_;
};

// ---- dataset.h --------------------------------------------------------------------------------------------------------------
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
    void SetTensorBeginId(int64_t b) { begin_id_ = b; }
    const size_t* GetLims() const { return lims_; }
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

inline DataSetPtr
GenDataSet(const int64_t nb, const int64_t dim, const void* xb, const int64_t beg_id = 0) {
    auto ds = std::make_shared<DataSet>();
    ds->SetRows(nb);
    ds->SetDim(dim);
    ds->SetTensor(xb);
    ds->SetTensorBeginId(beg_id);
    ds->SetIsOwner(false);
    return ds;
}
inline DataSetPtr
GenIdsDataSet(const int64_t rows, const int64_t* ids) {
    auto ds = std::make_shared<DataSet>();
    ds->SetRows(rows);
    ds->SetIds(ids);
    ds->SetIsOwner(false);
    return ds;
}
/// raw vectors result: takes ownership of a new[]-allocated tensor
inline DataSetPtr
GenResultDataSet(const int64_t rows, const int64_t dim, const void* tensor) {
    auto ds = std::make_shared<DataSet>();
    ds->SetRows(rows);
    ds->SetDim(dim);
    ds->SetTensor(tensor);
    ds->SetIsOwner(true);
    return ds;
}
/// takes ownership of two new[]-allocated arrays (include/knowhere/dataset.h:497-512)
inline DataSetPtr
GenResultDataSet(const int64_t nq, const int64_t topk, const int64_t* ids, const float* distance) {
    auto ds = std::make_shared<DataSet>();
    ds->SetRows(nq);
    ds->SetDim(topk);
    ds->SetIds(ids);
    ds->SetDistance(distance);
    ds->SetIsOwner(true);
    return ds;
}
/// range search result (include/knowhere/dataset.h:544): lims[nq + 1] + flat ids / distances, owned
inline DataSetPtr
GenResultDataSet(const int64_t nq, const int64_t* ids, const float* distance, const size_t* lims) {
    auto ds = std::make_shared<DataSet>();
    ds->SetRows(nq);
    ds->SetIds(ids);
    ds->SetDistance(distance);
    ds->SetLims(lims);
    ds->SetIsOwner(true);
    return ds;
}

// ---- binaryset.h ---------------------------------------------------------------------------------------------------------------
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

class Interrupt;

// ---- index_node.h ------------------------------------------------------------------------------------------------------------------
class IndexNode : public Object {
 public:
    IndexNode() = default;
    virtual ~IndexNode() = default;

    virtual Status
    Build(const DataSetPtr dataset, std::shared_ptr<Config> cfg, bool use_knowhere_build_pool = true) {
        RETURN_IF_ERROR(Train(dataset, cfg, use_knowhere_build_pool));
        return Add(dataset, std::move(cfg), use_knowhere_build_pool);
    }
    virtual Status
    BuildAsync(const DataSetPtr dataset, std::shared_ptr<Config> cfg, const Interrupt* = nullptr) {
        return Build(dataset, std::move(cfg), true);
    }
    virtual Status
    Train(const DataSetPtr dataset, std::shared_ptr<Config> cfg, bool use_knowhere_build_pool = true) = 0;
    virtual Status
    Add(const DataSetPtr dataset, std::shared_ptr<Config> cfg, bool use_knowhere_build_pool = true) = 0;
    virtual expected<DataSetPtr>
    Search(const DataSetPtr dataset, std::unique_ptr<Config> cfg, const BitsetView& bitset,
           milvus::OpContext* op_context = nullptr) const = 0;
    virtual expected<DataSetPtr>
    RangeSearch(const DataSetPtr dataset, std::unique_ptr<Config> cfg, const BitsetView& bitset,
                milvus::OpContext* op_context = nullptr) const {
        return expected<DataSetPtr>::Err(Status::not_implemented, "RangeSearch not implemented");
    }
    virtual expected<DataSetPtr>
    GetVectorByIds(const DataSetPtr dataset, milvus::OpContext* op_context = nullptr) const = 0;
    virtual bool
    HasRawData(const std::string& metric_type) const = 0;
    virtual bool
    NeedBitsetExactCount() const {
        return false;
    }
    virtual expected<DataSetPtr>
    GetIndexMeta(std::unique_ptr<Config> cfg) const = 0;
    virtual Status
    Serialize(BinarySet& binset) const = 0;
    virtual Status
    Deserialize(const BinarySet& binset, std::shared_ptr<Config> config) = 0;
    virtual Status
    DeserializeFromFile(const std::string& filename, std::shared_ptr<Config> config) = 0;
    virtual std::unique_ptr<BaseConfig>
    CreateConfig() const = 0;
    virtual int64_t
    Dim() const = 0;
    virtual int64_t
    Size() const = 0;
    virtual int64_t
    Count() const = 0;
    virtual IdMap&
    GetIdMap() {
        return id_map_;
    }
    virtual const IdMap&
    GetIdMap() const {
        return id_map_;
    }
    virtual std::string
    Type() const = 0;

    // Projects a public-id bitset to the backend id domain (include/knowhere/index/index_node.h:292-318)
    virtual BitsetView
    PrepareBitset(BitsetView bitset) const {
        const auto& id_map = GetIdMap();
        if (bitset.num_bits() == 0 || bitset.data() == nullptr) return bitset;
        if (!id_map.Empty()) bitset.set_out_ids(id_map.InToOut(), (size_t)id_map.OutCount());
        if (NeedBitsetExactCount() || !bitset.has_known_count()) bitset.count_filtered_bits();
        return bitset;
    }

 protected:
    void
    MapSearchResultIdsToOutIds(const DataSetPtr& result) const {
        // Backend storage ids -> public result ids.
        if (result == nullptr || result->GetIds() == nullptr) return;
        auto* ids = const_cast<int64_t*>(result->GetIds());
        const auto* lims = result->GetLims();
        const auto rows = result->GetRows();
        const auto count = lims != nullptr ? lims[rows] : static_cast<size_t>(rows * result->GetDim());
        GetIdMap().MapInToOut(ids, count);
    }
    IdMap id_map_;
};

// ---- index_node_thread_pool_wrapper.h: bounds the searches in flight on the device -----------------------------------------------
class IndexNodeThreadPoolWrapper : public IndexNode {
 public:
    IndexNodeThreadPoolWrapper(std::unique_ptr<IndexNode> index_node, size_t pool_size)
        : index_node_(std::move(index_node)), slots_(pool_size ? pool_size : 1) {}

    Status Train(const DataSetPtr dataset, std::shared_ptr<Config> cfg, bool use_knowhere_build_pool) override {
        return index_node_->Train(dataset, std::move(cfg), use_knowhere_build_pool);
    }
    Status Add(const DataSetPtr dataset, std::shared_ptr<Config> cfg, bool use_knowhere_build_pool) override {
        return index_node_->Add(dataset, std::move(cfg), use_knowhere_build_pool);
    }
    expected<DataSetPtr> Search(const DataSetPtr dataset, std::unique_ptr<Config> cfg, const BitsetView& bitset,
                                milvus::OpContext* op_context) const override {
        Slot s(this);
        return index_node_->Search(dataset, std::move(cfg), bitset, op_context);
    }
    expected<DataSetPtr> RangeSearch(const DataSetPtr dataset, std::unique_ptr<Config> cfg, const BitsetView& bitset,
                                     milvus::OpContext* op_context) const override {
        Slot s(this);
        return index_node_->RangeSearch(dataset, std::move(cfg), bitset, op_context);
    }
    expected<DataSetPtr> GetVectorByIds(const DataSetPtr dataset, milvus::OpContext* op_context) const override {
        return index_node_->GetVectorByIds(dataset, op_context);
    }
    bool HasRawData(const std::string& metric_type) const override { return index_node_->HasRawData(metric_type); }
    bool NeedBitsetExactCount() const override { return index_node_->NeedBitsetExactCount(); }
    expected<DataSetPtr> GetIndexMeta(std::unique_ptr<Config> cfg) const override {
        return index_node_->GetIndexMeta(std::move(cfg));
    }
    Status Serialize(BinarySet& binset) const override { return index_node_->Serialize(binset); }
    Status Deserialize(const BinarySet& binset, std::shared_ptr<Config> config) override {
        return index_node_->Deserialize(binset, std::move(config));
    }
    Status DeserializeFromFile(const std::string& filename, std::shared_ptr<Config> config) override {
        return index_node_->DeserializeFromFile(filename, std::move(config));
    }
    std::unique_ptr<BaseConfig> CreateConfig() const override { return index_node_->CreateConfig(); }
    int64_t Dim() const override { return index_node_->Dim(); }
    int64_t Size() const override { return index_node_->Size(); }
    int64_t Count() const override { return index_node_->Count(); }
    IdMap& GetIdMap() override { return index_node_->GetIdMap(); }
    const IdMap& GetIdMap() const override { return index_node_->GetIdMap(); }
    std::string Type() const override { return index_node_->Type(); }
    size_t PoolSize() const { return slots_; }
    size_t MaxInFlightSeen() const { return max_seen_; }

 private:
    struct Slot {
        const IndexNodeThreadPoolWrapper* w;
        explicit Slot(const IndexNodeThreadPoolWrapper* w_) : w(w_) {
            std::unique_lock<std::mutex> lk(w->mu_);
            w->cv_.wait(lk, [&] { return w->in_flight_ < w->slots_; });
            w->in_flight_++;
            w->max_seen_ = std::max(w->max_seen_, w->in_flight_);
        }
        ~Slot() {
            {
                std::lock_guard<std::mutex> lk(w->mu_);
                w->in_flight_--;
            }
            w->cv_.notify_one();
        }
    };
    std::unique_ptr<IndexNode> index_node_;
    size_t slots_;
    mutable std::mutex mu_;
    mutable std::condition_variable cv_;
    mutable size_t in_flight_ = 0, max_seen_ = 0;
};

// ---- index.h: ref-counted facade; every call is guarded (exceptions -> Status), as GuardedCall does -----------------------------
template <typename T1>
class Index {
 public:
    Index() = default;
    explicit Index(T1* node) : node_(node) {}
    template <typename... Args>
    static Index<T1>
    Create(Args&&... args) {
        return Index(new (std::nothrow) T1(std::forward<Args>(args)...));
    }
    template <typename T2>
    Index(const Index<T2>& other) : node_(other.NodePtr()) {}

    // json -> typed config of the node (src/index/index.cc LoadConfig): FormatAndCheck, then Load for the call's type
    static Status
    LoadConfig(BaseConfig* cfg, const Json& json, PARAM_TYPE type, std::string* msg) {
        Json j = json;
        RETURN_IF_ERROR(Config::FormatAndCheck(*cfg, j, msg));
        return Config::Load(*cfg, j, type, msg);
    }
    template <class F>
    static Status
    Guard(F&& f) noexcept {
        try {
            return f();
        } catch (const std::bad_alloc&) {
            return Status::malloc_error;
        } catch (const OperationCancelled&) {
            return Status::timeout;
        } catch (...) {
            return Status::knowhere_inner_error;
        }
    }
    Status
    Build(const DataSetPtr dataset, const Json& json, bool use_knowhere_build_pool = true) noexcept {
        return Guard([&] {
            std::shared_ptr<Config> cfg = node_->CreateConfig();
            std::string msg;
            RETURN_IF_ERROR(LoadConfig(static_cast<BaseConfig*>(cfg.get()), json, PARAM_TYPE::TRAIN, &msg));
            return node_->Build(dataset, cfg, use_knowhere_build_pool);
        });
    }
    Status
    Train(const DataSetPtr dataset, const Json& json, bool use_knowhere_build_pool = true) noexcept {
        return Guard([&] {
            std::shared_ptr<Config> cfg = node_->CreateConfig();
            std::string msg;
            RETURN_IF_ERROR(LoadConfig(static_cast<BaseConfig*>(cfg.get()), json, PARAM_TYPE::TRAIN, &msg));
            return node_->Train(dataset, cfg, use_knowhere_build_pool);
        });
    }
    Status
    Add(const DataSetPtr dataset, const Json& json, bool use_knowhere_build_pool = true) noexcept {
        return Guard([&] {
            std::shared_ptr<Config> cfg = node_->CreateConfig();
            std::string msg;
            RETURN_IF_ERROR(LoadConfig(static_cast<BaseConfig*>(cfg.get()), json, PARAM_TYPE::TRAIN, &msg));
            return node_->Add(dataset, cfg, use_knowhere_build_pool);
        });
    }
    expected<DataSetPtr>
    Search(const DataSetPtr dataset, const Json& json, const BitsetView& bitset,
           milvus::OpContext* op_context = nullptr) const noexcept {
        try {
            auto cfg = node_->CreateConfig();
            std::string msg;
            const Status s = LoadConfig(cfg.get(), json, PARAM_TYPE::SEARCH, &msg);
            if (s != Status::success) return expected<DataSetPtr>::Err(s, msg);
            return node_->Search(dataset, std::move(cfg), node_->PrepareBitset(bitset), op_context);
        } catch (const OperationCancelled& e) {
            return expected<DataSetPtr>::Err(Status::timeout, e.what());
        } catch (const std::exception& e) {
            return expected<DataSetPtr>::Err(Status::knowhere_inner_error, e.what());
        }
    }
    expected<DataSetPtr>
    RangeSearch(const DataSetPtr dataset, const Json& json, const BitsetView& bitset,
                milvus::OpContext* op_context = nullptr) const noexcept {
        try {
            auto cfg = node_->CreateConfig();
            std::string msg;
            const Status s = LoadConfig(cfg.get(), json, PARAM_TYPE::RANGE_SEARCH, &msg);
            if (s != Status::success) return expected<DataSetPtr>::Err(s, msg);
            return node_->RangeSearch(dataset, std::move(cfg), node_->PrepareBitset(bitset), op_context);
        } catch (const std::exception& e) {
            return expected<DataSetPtr>::Err(Status::knowhere_inner_error, e.what());
        }
    }
    expected<DataSetPtr>
    GetVectorByIds(const DataSetPtr dataset, milvus::OpContext* op_context = nullptr) const noexcept {
        try {
            return node_->GetVectorByIds(dataset, op_context);
        } catch (const std::exception& e) {
            return expected<DataSetPtr>::Err(Status::knowhere_inner_error, e.what());
        }
    }
    bool HasRawData(const std::string& metric_type) const { return node_->HasRawData(metric_type); }
    Status
    Serialize(BinarySet& binset) const noexcept {
        return Guard([&] { return node_->Serialize(binset); });
    }
    Status
    Deserialize(const BinarySet& binset, const Json& json = {}) noexcept {
        return Guard([&] {
            std::shared_ptr<Config> cfg = node_->CreateConfig();
            std::string msg;
            RETURN_IF_ERROR(LoadConfig(static_cast<BaseConfig*>(cfg.get()), json, PARAM_TYPE::DESERIALIZE, &msg));
            return node_->Deserialize(binset, cfg);
        });
    }
    // index.h:220 / src/index/index.cc:462-495
    Status
    DeserializeFromFile(const std::string& filename, const Json& json = {}) noexcept {
        return Guard([&] {
            std::shared_ptr<Config> cfg = node_->CreateConfig();
            std::string msg;
            RETURN_IF_ERROR(LoadConfig(static_cast<BaseConfig*>(cfg.get()), json, PARAM_TYPE::DESERIALIZE, &msg));
            return node_->DeserializeFromFile(filename, cfg);
        });
    }
    int64_t Dim() const { return node_->Dim(); }
    int64_t Size() const { return node_->Size(); }
    int64_t Count() const { return node_->Count(); }
    std::string Type() const { return node_->Type(); }
    T1* Node() const { return node_.get(); }
    std::shared_ptr<T1> NodePtr() const { return node_; }

 private:
    std::shared_ptr<T1> node_;
};

// ---- index_static.h ------------------------------------------------------------------------------------------------------------------
template <typename DataType>
class IndexStaticFaced {
 public:
    static std::unique_ptr<BaseConfig>
    CreateConfig(const knowhere::IndexType& indexType, const knowhere::IndexVersion& version) noexcept {
        auto& m = Instance().staticCreateConfigMap;
        auto it = m.find(indexType);
        return it == m.end() ? nullptr : it->second();
    }
    static knowhere::Status
    ConfigCheck(const knowhere::IndexType& indexType, const knowhere::IndexVersion& version, const knowhere::Json& params,
                std::string& msg) noexcept {
        auto cfg = CreateConfig(indexType, version);
        if (!cfg) {
            msg = "index type " + indexType + " is not registered";
            return Status::invalid_index_error;
        }
        Json j = params;
        Status s = Config::FormatAndCheck(*cfg, j, &msg);
        if (s != Status::success) return s;
        s = Config::Load(*cfg, j, PARAM_TYPE::TRAIN, &msg);
        if (s != Status::success) return s;
        auto& m = Instance().staticConfigCheckMap;
        auto it = m.find(indexType);
        return it == m.end() ? Status::success : it->second(*cfg, PARAM_TYPE::TRAIN, msg);
    }
    static bool
    HasRawData(const knowhere::IndexType& indexType, const knowhere::IndexVersion& version,
               const knowhere::Json& params) noexcept {
        auto cfg = CreateConfig(indexType, version);
        auto& m = Instance().staticHasRawDataMap;
        auto it = m.find(indexType);
        if (!cfg || it == m.end()) return false;
        Json j = params;
        std::string msg;
        if (Config::FormatAndCheck(*cfg, j, &msg) != Status::success) return false;
        if (Config::Load(*cfg, j, PARAM_TYPE::STATIC, &msg) != Status::success) return false;
        return it->second(*cfg, version);
    }
    template <typename VecIndexNode>
    IndexStaticFaced&
    RegisterStaticFunc(const knowhere::IndexType& indexType) {
        staticCreateConfigMap[indexType] = VecIndexNode::StaticCreateConfig;
        staticHasRawDataMap[indexType] = VecIndexNode::StaticHasRawData;
        staticConfigCheckMap[indexType] = VecIndexNode::StaticConfigCheck;
        return Instance();
    }
    static IndexStaticFaced&
    Instance() {
        static IndexStaticFaced f;
        return f;
    }

 private:
    std::map<std::string, std::function<std::unique_ptr<BaseConfig>()>> staticCreateConfigMap;
    std::map<std::string, std::function<bool(const knowhere::BaseConfig&, const IndexVersion&)>> staticHasRawDataMap;
    std::map<std::string, std::function<knowhere::Status(const knowhere::BaseConfig&, PARAM_TYPE, std::string&)>>
        staticConfigCheckMap;
};

// ---- index_factory.h -------------------------------------------------------------------------------------------------------------------
class IndexFactory {
 public:
    template <typename DataType>
    expected<Index<IndexNode>>
    Create(const std::string& name, const int32_t& version, const Object& object = nullptr) {
        std::lock_guard<std::mutex> lk(mu_);
        auto it = map_.find(Key<DataType>(name));
        if (it == map_.end()) {
            return expected<Index<IndexNode>>::Err(Status::invalid_index_error, "failed to find index " + name);
        }
        return it->second(version, object);
    }
    template <typename DataType>
    const IndexFactory&
    Register(const std::string& name, std::function<Index<IndexNode>(const int32_t&, const Object&)> func,
             const uint64_t features) {
        std::lock_guard<std::mutex> lk(mu_);
        map_[Key<DataType>(name)] = std::move(func);
        features_[name] = features;
        return *this;
    }
    bool
    FeatureCheck(const std::string& name, uint64_t feature) const {
        auto it = features_.find(name);
        return it != features_.end() && (it->second & feature) == feature;
    }
    static IndexFactory&
    Instance() {
        static IndexFactory f;
        return f;
    }

 private:
    template <typename DataType>
    static std::string
    Key(const std::string& name) {
        return name + "#" + typeid(DataType).name();
    }
    std::mutex mu_;
    std::map<std::string, std::function<Index<IndexNode>(const int32_t&, const Object&)>> map_;
    std::map<std::string, uint64_t> features_;
};

#define KNOWHERE_FACTOR_CONCAT(x, y) index_factory_ref_##x##y
#define KNOWHERE_REGISTER_GLOBAL(name, func, data_type, condition, features) \
    const IndexFactory& KNOWHERE_FACTOR_CONCAT(name, data_type) =            \
        condition ? IndexFactory::Instance().Register<data_type>(#name, func, features) : IndexFactory::Instance();

#define KNOWHERE_STATIC_CONCAT(x, y) index_static_ref_##x##y
#define KNOWHERE_REGISTER_STATIC(name, index_node, data_type, ...)               \
    const IndexStaticFaced<data_type>& KNOWHERE_STATIC_CONCAT(name, data_type) = \
        IndexStaticFaced<data_type>::Instance().RegisterStaticFunc<index_node<data_type, ##__VA_ARGS__>>(#name);

#define KNOWHERE_REGISTER_GLOBAL_WITH_THREAD_POOL(name, index_node, data_type, features, thread_size) \
    KNOWHERE_REGISTER_STATIC(name, index_node, data_type)                                             \
    KNOWHERE_REGISTER_GLOBAL(                                                                         \
        name,                                                                                         \
        [](const int32_t& version, const Object& object) {                                            \
            return (Index<IndexNodeThreadPoolWrapper>::Create(                                        \
                std::make_unique<index_node<data_type>>(version, object), thread_size));              \
        },                                                                                            \
        data_type, typeCheck<data_type>(features), features)

// ---- comp/brute_force.h ------------------------------------------------------------------------------------------------------------------
struct BruteForce {
    template <typename DataType>
    static expected<DataSetPtr>
    Search(const DataSetPtr base_dataset, const DataSetPtr query_dataset, const Json& config, const BitsetView& bitset,
           milvus::OpContext* op_context = nullptr);
    // (include/knowhere/comp/brute_force.h:41-58: the caller's result buffers / the radius search)
    template <typename DataType>
    static Status
    SearchWithBuf(const DataSetPtr base_dataset, const DataSetPtr query_dataset, int64_t* ids, float* dis, const Json& config,
                  const BitsetView& bitset, milvus::OpContext* op_context = nullptr);
    template <typename DataType>
    static expected<DataSetPtr>
    RangeSearch(const DataSetPtr base_dataset, const DataSetPtr query_dataset, const Json& config, const BitsetView& bitset,
                milvus::OpContext* op_context = nullptr);
};

}  // namespace knowhere
