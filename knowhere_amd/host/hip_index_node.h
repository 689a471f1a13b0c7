// knowhere_amd/host/hip_index_node.h -- index names and Config classes of the MI355X backend.
//
// Written against the REAL Knowhere interface: with -DKNHIP_WITH_KNOWHERE_HEADERS it includes the reference's own
// headers (this is how it is built inside a Knowhere tree, and how tests/test_node_contract.py compile-checks it against
// /root/reference/include); without it, knowhere_shim.h supplies the same names (same signatures) so the node also
// builds and runs where the reference tree and its third-party dependencies are absent.
#pragma once

#if defined(KNHIP_WITH_KNOWHERE_HEADERS)
#include "index/flat/flat_config.h"
#include "index/ivf/ivf_config.h"
#include "knowhere/config.h"
#include "knowhere/context.h"
#include "knowhere/index/index_factory.h"
#include "knowhere/index/index_node.h"
#include "knowhere/index/index_node_data_mock_wrapper.h"
#include "knowhere/index/index_node_thread_pool_wrapper.h"
#else
#include "knowhere_shim.h"
#endif

#include <cctype>
#include <cstdlib>
#include <string>
#include <vector>

namespace knowhere {

// new index types, next to INDEX_CUVS_* / INDEX_GPU_* (include/knowhere/comp/index_param.h:42-55); the same pairs go
// into the legal-index table (include/knowhere/index/index_table.h:73-84)
namespace IndexEnum {
inline constexpr const char* INDEX_HIP_BRUTEFORCE = "GPU_HIP_BRUTE_FORCE";
inline constexpr const char* INDEX_HIP_IVFFLAT = "GPU_HIP_IVF_FLAT";
inline constexpr const char* INDEX_HIP_IVFPQ = "GPU_HIP_IVF_PQ";
inline constexpr const char* INDEX_HIP_IVFSQ8 = "GPU_HIP_IVF_SQ8";
}  // namespace IndexEnum

// ---- configs: the CPU index's config + the limits of this backend, in the style of
// src/index/gpu_cuvs/gpu_cuvs_ivf_pq_config.h:27-95 (k <= 1024 as the cuVS configs, :49-53) -------------------------
inline Status
HipCheckMetric(const BaseConfig& cfg, PARAM_TYPE param_type, std::string* err_msg) {
    if (param_type == PARAM_TYPE::TRAIN && cfg.metric_type.has_value()) {
        const std::string& m = cfg.metric_type.value();
        if (!IsMetricType(m, metric::L2) && !IsMetricType(m, metric::IP) && !IsMetricType(m, metric::COSINE)) {
            if (err_msg) *err_msg = "metric type " + m + " not found or not supported, supported: [L2 IP COSINE]";
            return Status::invalid_metric_type;
        }
    }
    return Status::success;
}

// Device placement, the same keys in all four configs (the reference's GPU configs carry `gpu_id`,
// src/index/gpu/ivf_gpu/ivf_gpu_config.h:18-20; its cuVS nodes place every new index round-robin and a loaded one on the
// device with the most free memory, src/common/cuvs/integration/cuvs_knowhere_index.cuh:414-426, 678-690):
//   gpu_id   one device ordinal.                      unset = round-robin over the visible devices at Train, the device
//                                                     with the most free HBM at Deserialize -- as the cuVS nodes.
//   gpu_ids  "0,1,2,3" or "all": the inverted lists (FLAT: the rows) are dealt over these devices and every Search() is
//            served by all of them (include/knhip_shards.h: one all-gather of the per-device top-k over RCCL / xGMI).
//            The reference's config system has no list type (include/knowhere/config.h:37-58), hence a string.  A
//            device may be named twice ("0,0"): the shards then share it and exchange by device copies (tests on a
//            one-GPU box).
#define KNHIP_DEVICE_CONFIG_MEMBERS \
    CFG_INT gpu_id;                 \
    CFG_STRING gpu_ids;
#define KNHIP_DEVICE_CONFIG_FIELDS()                                                                        \
    KNOWHERE_CONFIG_DECLARE_FIELD(gpu_id)                                                                   \
        .description("device ordinal of the index (unset: round-robin / most free memory)")                \
        .allow_empty_without_default()                                                                      \
        .set_range(0, 1023)                                                                                 \
        .for_train()                                                                                        \
        .for_deserialize()                                                                                  \
        .for_deserialize_from_file();                                                                       \
    KNOWHERE_CONFIG_DECLARE_FIELD(gpu_ids)                                                                  \
        .description("devices the index is sharded over: comma separated ordinals or \"all\"")             \
        .allow_empty_without_default()                                                                      \
        .for_train()                                                                                        \
        .for_deserialize()                                                                                  \
        .for_deserialize_from_file()

// "0,2,3" / "all" -> ordinals; empty on a syntax error or an ordinal outside [0, ndev)
inline std::vector<int32_t>
HipParseGpuIds(const std::string& s, int ndev) {
    std::vector<int32_t> out;
    std::string t;
    for (char c : s) {
        if (!std::isspace((unsigned char)c)) t.push_back((char)std::tolower((unsigned char)c));
    }
    if (t == "all") {
        for (int i = 0; i < ndev; i++) out.push_back(i);
        return out;
    }
    size_t pos = 0;
    while (pos <= t.size()) {
        const size_t e = t.find(',', pos);
        const std::string tok = t.substr(pos, e == std::string::npos ? std::string::npos : e - pos);
        if (tok.empty() || tok.size() > 4 || tok.find_first_not_of("0123456789") != std::string::npos) return {};
        const int v = std::atoi(tok.c_str());
        if (v < 0 || v >= ndev) return {};
        out.push_back(v);
        if (e == std::string::npos) break;
        pos = e + 1;
    }
    return out;
}

struct HipBruteForceConfig : public FlatConfig {
    KNHIP_DEVICE_CONFIG_MEMBERS
    KNOWHERE_DECLARE_CONFIG(HipBruteForceConfig) {
        KNHIP_DEVICE_CONFIG_FIELDS();
        KNOWHERE_CONFIG_DECLARE_FIELD(k)
            .set_default(10)
            .description("search for top k similar vector.")
            .set_range(1, 1024)
            .for_search();
    }
    Status
    CheckAndAdjust(PARAM_TYPE param_type, std::string* err_msg) override {
        return HipCheckMetric(*this, param_type, err_msg);
    }
};

struct HipIvfFlatConfig : public IvfFlatConfig {
    KNHIP_DEVICE_CONFIG_MEMBERS
    KNOWHERE_DECLARE_CONFIG(HipIvfFlatConfig) {
        KNHIP_DEVICE_CONFIG_FIELDS();
        KNOWHERE_CONFIG_DECLARE_FIELD(k)
            .set_default(10)
            .description("search for top k similar vector.")
            .set_range(1, 1024)
            .for_search();
    }
    Status
    CheckAndAdjust(PARAM_TYPE param_type, std::string* err_msg) override {
        return HipCheckMetric(*this, param_type, err_msg);
    }
};

struct HipIvfPqConfig : public IvfPqConfig {
    KNHIP_DEVICE_CONFIG_MEMBERS
    KNOWHERE_DECLARE_CONFIG(HipIvfPqConfig) {
        KNHIP_DEVICE_CONFIG_FIELDS();
        KNOWHERE_CONFIG_DECLARE_FIELD(k)
            .set_default(10)
            .description("search for top k similar vector.")
            .set_range(1, 1024)
            .for_search();
        // m = 0: the backend picks (about dim / 2 sub-quantizers, as cuVS does for pq_dim = 0)
        KNOWHERE_CONFIG_DECLARE_FIELD(m).set_default(0).description("m").set_range(0, 65536).for_train();
        // codes of 1 .. 8 bits (the cuVS config accepts 4 .. 8, gpu_cuvs_ivf_pq_config.h:55-58; the CPU node up to 24): on the
        // device every width is one byte per sub-quantizer indexing 256-entry tables of which 2^nbits are in use
        KNOWHERE_CONFIG_DECLARE_FIELD(nbits).set_default(8).description("nbits").set_range(1, 8).for_train();
    }
    Status
    CheckAndAdjust(PARAM_TYPE param_type, std::string* err_msg) override {
        RETURN_IF_ERROR(HipCheckMetric(*this, param_type, err_msg));
        if (param_type == PARAM_TYPE::TRAIN && m.has_value() && m.value() != 0) {
            const int mv = m.value();
            // (the reference takes any m that divides dim, ivf_config.h:118, :138-147; here 8, 16, 32, 64 run on the fast
            // kernels and every other m up to 128 on the plain exact one, sub-vectors of at most 144 dimensions)
            if (mv < 0 || mv > 128) {
                if (err_msg) *err_msg = "GPU_HIP_IVF_PQ supports m in 0 (auto), 1 ... 128";
                return Status::invalid_args;
            }
            if (dim.has_value() && dim.value() % mv != 0) {
                if (err_msg) *err_msg = "The dimension of a vector (dim) should be a multiple of the number of subquantizers (m)";
                return Status::invalid_args;
            }
            if (dim.has_value() && dim.value() / mv > 144) {
                if (err_msg) *err_msg = "GPU_HIP_IVF_PQ: sub-vectors (dim / m) of more than 144 dimensions are not supported";
                return Status::invalid_args;
            }
        }
        return Status::success;
    }
};

struct HipIvfSqConfig : public IvfSqConfig {
    KNHIP_DEVICE_CONFIG_MEMBERS
    KNOWHERE_DECLARE_CONFIG(HipIvfSqConfig) {
        KNHIP_DEVICE_CONFIG_FIELDS();
        KNOWHERE_CONFIG_DECLARE_FIELD(k)
            .set_default(10)
            .description("search for top k similar vector.")
            .set_range(1, 1024)
            .for_search();
    }
    Status
    CheckAndAdjust(PARAM_TYPE param_type, std::string* err_msg) override {
        RETURN_IF_ERROR(HipCheckMetric(*this, param_type, err_msg));
        if (param_type == PARAM_TYPE::TRAIN && sq_type.has_value()) {
            std::string t = sq_type.value();
            for (auto& c : t) c = (char)std::toupper((unsigned char)c);
            if (t != "SQ8") {
                if (err_msg) *err_msg = "GPU_HIP_IVF_SQ8 supports sq_type SQ8 only";
                return Status::invalid_args;
            }
        }
        return Status::success;
    }
};

// bounded in-flight searches per device, as the cuVS nodes (src/index/gpu_cuvs/gpu_cuvs.h:48)
auto static constexpr hip_concurrent_size_per_device = std::uint32_t{4};
size_t HipSearchPoolSize();

}  // namespace knowhere
