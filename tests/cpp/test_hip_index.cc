// tests/cpp/test_hip_index.cc -- the reference's GPU search tests, re-run against the HIP node through
// the Knowhere plugin API (IndexFactory::Create -> Index::Build/Search/Serialize/Deserialize).
//
// Mirrors reference tests/ut/test_gpu_search.cc:60-330 (same sizes nb=10000 nq=1000 dim=128, seeds
// 42 / 44, same generators and the same bars: self-search ids[i]==i, recall vs BruteForce::Search
// >= 0.999 / 0.95 / 0.75 for brute force / IVF-Flat / IVF-PQ, bitset at 40 % and 98 % filtered with
// recall floors 0.7 / 0.4, k in {5, 25, 100}, serialize round trip) and tests/ut/test_bruteforce.cc
// :70-76 (self-hit distance exactly 0).  Catch2 is absent from this image, hence the tiny harness.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <random>
#include <set>

#include "../../knowhere_amd/host/knowhere_shim.h"

static int g_fail = 0, g_checks = 0;
#define REQUIRE(cond)                                                              \
    do {                                                                           \
        g_checks++;                                                                \
        if (!(cond)) {                                                             \
            g_fail++;                                                              \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);            \
        }                                                                          \
    } while (0)

// tests/ut/utils.h:41-50
static knowhere::DataSetPtr GenDataSet(int rows, int dim, int seed = 42) {
    std::mt19937 rng(seed);
    std::uniform_real_distribution<> distrib(0.0, 100.0);
    float* ts = new float[(size_t)rows * dim];
    for (size_t i = 0; i < (size_t)rows * dim; ++i) ts[i] = distrib(rng);
    auto ds = knowhere::GenDataSet(rows, dim, ts);
    ds->SetIsOwner(true);
    return ds;
}

// tests/ut/utils.h:110-134
static float GetKNNRecall(const knowhere::DataSet& gt, const knowhere::DataSet& res) {
    auto nq = res.GetRows();
    auto gt_k = gt.GetDim();
    auto res_k = res.GetDim();
    uint32_t matched = 0;
    for (int64_t i = 0; i < nq; ++i) {
        std::set<int64_t> a(gt.GetIds() + i * gt_k, gt.GetIds() + i * gt_k + res_k);
        for (int64_t j = 0; j < res_k; j++) matched += a.count(res.GetIds()[i * res_k + j]);
    }
    return (float)matched / ((float)nq * res_k);
}

// tests/ut/utils.h GenerateBitsetWithFirstTbitsSet / RandomTbitsSet
static std::vector<uint8_t> BitsetFirst(size_t n, size_t t) {
    std::vector<uint8_t> b((n + 7) / 8, 0);
    for (size_t i = 0; i < t; i++) b[i >> 3] |= (1 << (i & 7));
    return b;
}
static std::vector<uint8_t> BitsetRandom(size_t n, size_t t) {
    std::vector<size_t> idx(n);
    for (size_t i = 0; i < n; i++) idx[i] = i;
    std::mt19937 rng(7);
    std::shuffle(idx.begin(), idx.end(), rng);
    std::vector<uint8_t> b((n + 7) / 8, 0);
    for (size_t i = 0; i < t; i++) b[idx[i] >> 3] |= (1 << (idx[i] & 7));
    return b;
}

int main() {
    using namespace knowhere;
    const int64_t nb = 10000, nq = 1000, dim = 128, seed = 42;
    auto version = Version::GetCurrentVersion();
    auto base_gen = [=]() {
        Json json;
        json[meta::DIM] = dim;
        json[meta::METRIC_TYPE] = metric::L2;
        json[meta::TOPK] = 1;
        return json;
    };
    auto ivfflat_gen = [=]() {
        Json json = base_gen();
        json[indexparam::NLIST] = 20;
        json[indexparam::NPROBE] = 18;
        return json;
    };
    auto ivfpq_gen = [=]() {
        Json json = ivfflat_gen();
        json[indexparam::M] = 0;
        json[indexparam::NBITS] = 8;
        return json;
    };
    struct Case {
        const char* name;
        Json cfg;
        float min_recall;
    };
    std::vector<Case> cases = {{IndexEnum::INDEX_HIP_BRUTEFORCE, base_gen(), 0.999f},
                               {IndexEnum::INDEX_HIP_IVFFLAT, ivfflat_gen(), 0.95f},
                               {IndexEnum::INDEX_HIP_IVFPQ, ivfpq_gen(), 0.75f},
                               {IndexEnum::INDEX_HIP_IVFSQ8, ivfflat_gen(), 0.95f}};

    auto train_ds = GenDataSet(nb, dim, seed);
    auto query_ds = GenDataSet(nq, dim, seed + 2);

    REQUIRE(!IndexFactory::Instance().Create<fp32>("NO_SUCH_INDEX", version).has_value());
    REQUIRE(IndexFactory::Instance().Create<fp32>("NO_SUCH_INDEX", version).error() == Status::invalid_index_error);

    // range search check (tests/ut/test_search.cc range sections): every result inside [range_filter, radius), and
    // with the early stop off the top-1 neighbour is never missed
    auto check_range = [&](Index<IndexNode>& idx, const Json& base_cfg) {
        Json rcfg = base_cfg;
        auto top = idx.Search(query_ds, rcfg, nullptr);  // k = 1 distances give a sensible radius
        REQUIRE(top.has_value());
        std::vector<float> d1(top.value()->GetDistance(), top.value()->GetDistance() + nq);
        std::nth_element(d1.begin(), d1.begin() + nq / 2, d1.end());
        const float radius = d1[nq / 2] * 1.5f;
        rcfg[meta::RADIUS] = radius;
        rcfg[meta::RANGE_FILTER] = 0.0f;
        rcfg[indexparam::MAX_EMPTY_RESULT_BUCKETS] = 0;
        auto rr = idx.RangeSearch(query_ds, rcfg, nullptr);
        REQUIRE(rr.has_value());
        if (rr.has_value()) {
            const size_t* lims = rr.value()->GetLims();
            int out_of_range = 0, missed = 0;
            for (int64_t i = 0; i < nq; i++) {
                bool has_top = false;
                for (size_t j = lims[i]; j < lims[i + 1]; j++) {
                    const float v = rr.value()->GetDistance()[j];
                    out_of_range += !(v >= 0.0f && v < radius);
                    has_top |= rr.value()->GetIds()[j] == top.value()->GetIds()[i];
                }
                missed += (top.value()->GetDistance()[i] < radius) && !has_top;
            }
            REQUIRE(lims[nq] > 0);
            REQUIRE(out_of_range == 0);
            REQUIRE(missed == 0);
            std::printf("   range search: %zu results inside radius %.3f\n", lims[nq], radius);
        }
        Json bad = rcfg;
        bad[meta::RADIUS] = "far";
        REQUIRE(idx.RangeSearch(query_ds, bad, nullptr).error() == Status::type_conflict_in_json);
    };

    for (auto& c : cases) {
        std::printf("== %s\n", c.name);
        auto idx = IndexFactory::Instance().Create<fp32>(c.name, version).value();
        REQUIRE(idx.Type() == c.name);
        // searching before Build: empty_index
        REQUIRE(idx.Search(query_ds, c.cfg, nullptr).error() == Status::empty_index);
        // 1. self-search (test_gpu_search.cc:78-86)
        REQUIRE(idx.Build(train_ds, c.cfg) == Status::success);
        REQUIRE(idx.Count() == nb);
        REQUIRE(idx.Dim() == dim);
        REQUIRE(idx.Size() > 0);
        auto results = idx.Search(train_ds, c.cfg, nullptr);
        REQUIRE(results.has_value());
        if (results.has_value()) {
            auto ids = results.value()->GetIds();
            int bad = 0;
            for (int i = 1; i < nq; ++i) bad += ids[i] != i;
            // exact for brute force / flat / sq8; PQ codes can collide
            REQUIRE(bad <= (std::string(c.name) == IndexEnum::INDEX_HIP_IVFPQ ? nq / 50 : 0));
            if (std::string(c.name) == IndexEnum::INDEX_HIP_BRUTEFORCE ||
                std::string(c.name) == IndexEnum::INDEX_HIP_IVFFLAT) {
                int nz = 0;  // tests/ut/test_bruteforce.cc:70-76: L2 self distance exactly 0
                for (int i = 0; i < nq; ++i) nz += results.value()->GetDistance()[i] != 0.0f;
                REQUIRE(nz == 0);
            }
        }
        // 2. recall vs BruteForce::Search (test_gpu_search.cc:90-97)
        results = idx.Search(query_ds, c.cfg, nullptr);
        REQUIRE(results.has_value());
        auto gt = BruteForce::Search<fp32>(train_ds, query_ds, c.cfg, nullptr);
        REQUIRE(gt.has_value());
        float recall = GetKNNRecall(*gt.value(), *results.value());
        std::printf("   recall@1 %.4f (floor %.3f)\n", recall, c.min_recall);
        REQUIRE(recall >= c.min_recall);
        // 3. larger k (test_gpu_search.cc:245-278)
        for (int k : {5, 25, 100}) {
            Json cfg = c.cfg;
            cfg[meta::TOPK] = k;
            auto r = idx.Search(query_ds, cfg, nullptr);
            auto g = BruteForce::Search<fp32>(train_ds, query_ds, cfg, nullptr);
            REQUIRE(r.has_value() && g.has_value());
            float rc = GetKNNRecall(*g.value(), *r.value());
            std::printf("   recall@%d %.4f\n", k, rc);
            REQUIRE(rc >= (std::string(c.name) == IndexEnum::INDEX_HIP_IVFPQ ? 0.5f : c.min_recall - 0.05f));
        }
        // 4. bitset 40 % / 98 % filtered, two bit patterns (test_gpu_search.cc:204-243)
        for (float frac : {0.4f, 0.98f}) {
            for (int pat = 0; pat < 2; pat++) {
                auto bits = pat == 0 ? BitsetFirst(nb, (size_t)(frac * nb)) : BitsetRandom(nb, (size_t)(frac * nb));
                BitsetView bv(bits.data(), nb);
                auto r = idx.Search(query_ds, c.cfg, bv);
                auto g = BruteForce::Search<fp32>(train_ds, query_ds, c.cfg, bv);
                REQUIRE(r.has_value() && g.has_value());
                int leaked = 0;
                for (int64_t i = 0; i < nq; i++) {
                    int64_t id = r.value()->GetIds()[i];
                    leaked += id >= 0 && bv.test(id);
                }
                REQUIRE(leaked == 0);
                float rc = GetKNNRecall(*g.value(), *r.value());
                REQUIRE(rc > (frac < 0.5f ? 0.7f : 0.4f));
            }
        }
        {   // everything filtered: ids -1 (gpu_cuvs.h:163-173)
            auto bits = BitsetFirst(nb, nb);
            auto r = idx.Search(query_ds, c.cfg, BitsetView(bits.data(), nb));
            REQUIRE(r.has_value() && r.value()->GetIds()[0] == -1);
        }
        // 5. refine (IVF_PQ / IVF_SQ8): a build-time `refine` keeps the fp32 rows (ivf.cc:673-700), a
        //    search-time `refine_k` re-ranks with them (ivf.cc:1073-1103); without the former the latter
        //    is a no-op
        if (std::string(c.name) == IndexEnum::INDEX_HIP_IVFPQ) {
            Json cfg = c.cfg;
            cfg[meta::TOPK] = 10;
            auto plain = idx.Search(query_ds, cfg, nullptr);
            Json scfg = cfg;
            scfg[indexparam::REFINE_K] = 100;
            auto noop = idx.Search(query_ds, scfg, nullptr);
            REQUIRE(plain.has_value() && noop.has_value());
            int diff = 0;
            for (int64_t i = 0; i < nq * 10; i++) diff += plain.value()->GetIds()[i] != noop.value()->GetIds()[i];
            REQUIRE(diff == 0);
            Json bcfg = c.cfg;
            bcfg[indexparam::REFINE] = true;
            auto ridx = IndexFactory::Instance().Create<fp32>(c.name, version).value();
            REQUIRE(ridx.Build(train_ds, bcfg) == Status::success);
            auto refined = ridx.Search(query_ds, scfg, nullptr);
            auto g = BruteForce::Search<fp32>(train_ds, query_ds, cfg, nullptr);
            REQUIRE(refined.has_value() && g.has_value());
            float r0 = GetKNNRecall(*g.value(), *plain.value()), r1 = GetKNNRecall(*g.value(), *refined.value());
            std::printf("   recall@10 plain %.4f refined(100) %.4f\n", r0, r1);
            REQUIRE(r1 > r0 && r1 > 0.9f);
            // the refine index travels with the blob ("IxRF" wrapper) and keeps working after a reload
            BinarySet rbs;
            REQUIRE(ridx.Serialize(rbs) == Status::success);
            REQUIRE(std::memcmp(rbs.GetByName(c.name)->data.get(), "IxRF", 4) == 0);
            auto ridx2 = IndexFactory::Instance().Create<fp32>(c.name, version).value();
            REQUIRE(ridx2.Deserialize(rbs) == Status::success);
            auto refined2 = ridx2.Search(query_ds, scfg, nullptr);
            REQUIRE(refined2.has_value());
            diff = 0;
            for (int64_t i = 0; i < nq * 10; i++) diff += refined.value()->GetIds()[i] != refined2.value()->GetIds()[i];
            REQUIRE(diff == 0);
        }
        // 6. serialize round trip (test_gpu_search.cc:280-314)
        BinarySet bs;
        REQUIRE(idx.Serialize(bs) == Status::success);
        REQUIRE(bs.Contains(c.name));
        {   // the blob is the FAISS byte format of the CPU node of the same kind (index_write.cpp fourccs)
            const char* cc = std::string(c.name) == IndexEnum::INDEX_HIP_BRUTEFORCE ? "IxF2"
                             : std::string(c.name) == IndexEnum::INDEX_HIP_IVFFLAT  ? "IwFl"
                             : std::string(c.name) == IndexEnum::INDEX_HIP_IVFPQ    ? "IwPQ" : "IwSq";
            REQUIRE(std::memcmp(bs.GetByName(c.name)->data.get(), cc, 4) == 0);
        }
        auto idx2 = IndexFactory::Instance().Create<fp32>(c.name, version).value();
        REQUIRE(idx2.Deserialize(bs) == Status::success);
        auto r2 = idx2.Search(query_ds, c.cfg, nullptr);
        REQUIRE(r2.has_value());
        if (r2.has_value()) {
            int diff = 0;
            for (int64_t i = 0; i < nq; i++) diff += r2.value()->GetIds()[i] != results.value()->GetIds()[i];
            REQUIRE(diff == 0);
        }
        // 7. range search: brute force, IVF_FLAT, IVF_SQ8 here, IVF_PQ (m = 32) below; other m: not_implemented
        if (std::string(c.name) == IndexEnum::INDEX_HIP_IVFPQ) {
            REQUIRE(idx.RangeSearch(query_ds, c.cfg, nullptr).error() == Status::not_implemented);
        } else {
            check_range(idx, c.cfg);
        }
        // config validation
        Json bad = c.cfg;
        bad[meta::TOPK] = 100000;
        REQUIRE(idx.Search(query_ds, bad, nullptr).error() == Status::out_of_range_in_json);
        bad = c.cfg;
        bad[meta::METRIC_TYPE] = "HAMMING";
        REQUIRE(idx.Search(query_ds, bad, nullptr).error() == Status::invalid_metric_type);
    }

    {   // IVF_PQ with m = 32: the range search path of the headline kernel
        Json cfg = ivfpq_gen();
        cfg[indexparam::M] = 32;
        auto idx = IndexFactory::Instance().Create<fp32>(IndexEnum::INDEX_HIP_IVFPQ, version).value();
        REQUIRE(idx.Build(train_ds, cfg) == Status::success);
        check_range(idx, cfg);
    }

    {   // COSINE == normalised IP
        Json cfg = ivfflat_gen();
        cfg[meta::METRIC_TYPE] = metric::COSINE;
        cfg[meta::TOPK] = 5;
        auto idx = IndexFactory::Instance().Create<fp32>(IndexEnum::INDEX_HIP_IVFFLAT, version).value();
        REQUIRE(idx.Build(train_ds, cfg) == Status::success);
        auto r = idx.Search(train_ds, cfg, nullptr);
        REQUIRE(r.has_value());
        int bad = 0;
        for (int i = 0; i < nq; i++) bad += r.value()->GetIds()[i * 5] != i;
        REQUIRE(bad == 0);
        REQUIRE(std::abs(r.value()->GetDistance()[0] - 1.0f) < 1e-5f);
    }

    std::printf("%s: %d checks, %d failed\n", g_fail ? "FAILED" : "PASSED", g_checks, g_fail);
    return g_fail ? 1 : 0;
}
