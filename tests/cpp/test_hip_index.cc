// tests/cpp/test_hip_index.cc -- the reference's GPU search tests, re-run against the HIP node through
// the Knowhere plugin API (IndexFactory::Create -> Index::Build/Search/Serialize/Deserialize).
//
// Mirrors reference tests/ut/test_gpu_search.cc:60-330 (same sizes nb=10000 nq=1000 dim=128, seeds
// 42 / 44, same generators and the same bars: self-search ids[i]==i, recall vs BruteForce::Search
// >= 0.999 / 0.95 / 0.75 for brute force / IVF-Flat / IVF-PQ, bitset at 40 % and 98 % filtered with
// recall floors 0.7 / 0.4, k in {5, 25, 100}, serialize round trip) and tests/ut/test_bruteforce.cc
// :70-76 (self-hit distance exactly 0); plus the node contract: static functions (index_static.h), the thread-pool
// wrapper, id-map out ids, repeated Add, GetVectorByIds, cancellation, Deserialize of damaged blobs.
// Catch2 is absent from this image, hence the tiny harness.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <set>
#include <thread>

#include "hip_index_node.h"

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

// An exhaustive scan on the HOST, in double precision, written here: the ground truth of the ground truth.  The recall bars
// below compare the IVF indexes with BruteForce::Search, which in the standalone build is served by the device as well
// (host/hip_brute_force.cc); this check ties that device scan to arithmetic that never touches the GPU.  Returns the number
// of (query, rank) places where the device result is not explained by the host scan: the id must be among the host's rows
// whose distance lies within `tol` (relative) of the host's value at that rank, and the distance must agree with the id's own.
static int HostScanMismatches(const knowhere::DataSet& base, const knowhere::DataSet& queries, const knowhere::DataSet& res,
                              bool is_l2, const uint8_t* bitset = nullptr, double tol = 5e-5) {
    const int64_t nb = base.GetRows(), d = base.GetDim(), nq = queries.GetRows(), k = res.GetDim();
    const float* xb = (const float*)base.GetTensor();
    const float* xq = (const float*)queries.GetTensor();
    int bad = 0;
    std::vector<std::pair<double, int64_t>> all;
    for (int64_t q = 0; q < nq; q++) {
        all.clear();
        for (int64_t i = 0; i < nb; i++) {
            if (bitset != nullptr && (bitset[i >> 3] >> (i & 7)) & 1) continue;
            double acc = 0.0;
            for (int64_t j = 0; j < d; j++) {
                const double a = xq[q * d + j], b = xb[i * d + j];
                acc += is_l2 ? (a - b) * (a - b) : a * b;
            }
            all.emplace_back(is_l2 ? acc : -acc, i);  // ascending = best first for both metrics
        }
        std::sort(all.begin(), all.end());
        std::vector<double> of_id(nb, 1e300);
        for (auto& e : all) of_id[e.second] = e.first;
        for (int64_t r = 0; r < k; r++) {
            const int64_t id = res.GetIds()[q * k + r];
            if (r >= (int64_t)all.size()) {
                bad += id != -1;
                continue;
            }
            const double want = all[r].first, scale = std::max(1.0, std::fabs(want));
            if (id < 0 || id >= nb || std::fabs(of_id[id] - want) > tol * scale) {
                bad++;
                continue;
            }
            const double got = is_l2 ? res.GetDistance()[q * k + r] : -(double)res.GetDistance()[q * k + r];
            bad += std::fabs(got - of_id[id]) > tol * scale;
        }
    }
    return bad;
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
    auto version = Version::GetCurrentVersion().VersionNumber();
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
        REQUIRE(idx.RangeSearch(query_ds, bad, nullptr).error() == Status::invalid_value_in_json);
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
        {   // the ground truth itself against a host scan in double precision (not the device's own word for it)
            const int bad = HostScanMismatches(*train_ds, *query_ds, *gt.value(), /*is_l2=*/true);
            std::printf("   BruteForce::Search vs host double-precision scan: %d places differ\n", bad);
            REQUIRE(bad == 0);
        }
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
            if (std::string(c.name) == IndexEnum::INDEX_HIP_BRUTEFORCE) {
                REQUIRE(HostScanMismatches(*train_ds, *query_ds, *g.value(), true) == 0);
            }
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
                if (std::string(c.name) == IndexEnum::INDEX_HIP_BRUTEFORCE) {  // (once: the same call for every index)
                    REQUIRE(HostScanMismatches(*train_ds, *query_ds, *g.value(), true, bits.data()) == 0);
                }
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
            scfg[indexparam::REFINE_K] = 10.0f;  // a k factor: k_base = k * refine_k (ivf.cc:1081)
            auto noop = idx.Search(query_ds, scfg, nullptr);
            REQUIRE(plain.has_value() && noop.has_value());
            int diff = 0;
            for (int64_t i = 0; i < nq * 10; i++) diff += plain.value()->GetIds()[i] != noop.value()->GetIds()[i];
            REQUIRE(diff == 0);
            auto g = BruteForce::Search<fp32>(train_ds, query_ds, cfg, nullptr);
            REQUIRE(g.has_value());
            const float r0 = GetKNNRecall(*g.value(), *plain.value());
            // refine_type: fp32 / flat = IndexRefineFlat, fp16 / bf16 / sq8 = IndexRefine over an IndexScalarQuantizer
            // (refine_utils.cc:99-185); both `refine` and `refine_type` are needed for a refine index (ivf_wrapper.cc:170)
            for (const char* rtype : {"fp32", "fp16", "bf16", "sq8"}) {
                Json bcfg = c.cfg;
                bcfg[indexparam::REFINE] = true;
                bcfg[indexparam::REFINE_TYPE] = rtype;
                auto ridx = IndexFactory::Instance().Create<fp32>(c.name, version).value();
                REQUIRE(ridx.Build(train_ds, bcfg) == Status::success);
                auto refined = ridx.Search(query_ds, scfg, nullptr);
                REQUIRE(refined.has_value());
                const float r1 = GetKNNRecall(*g.value(), *refined.value());
                std::printf("   recall@10 plain %.4f refined(k x 10, %s) %.4f\n", r0, rtype, r1);
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
            {   // `refine` alone builds no refine index
                Json bcfg = c.cfg;
                bcfg[indexparam::REFINE] = true;
                auto nidx = IndexFactory::Instance().Create<fp32>(c.name, version).value();
                REQUIRE(nidx.Build(train_ds, bcfg) == Status::success);
                BinarySet nbs;
                REQUIRE(nidx.Serialize(nbs) == Status::success);
                REQUIRE(std::memcmp(nbs.GetByName(c.name)->data.get(), "IwPQ", 4) == 0);
            }
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
        {   // 6b. the same bytes from a file (Index::DeserializeFromFile, src/index/index.cc:462-495 ->
            //     IvfIndexNode::DeserializeFromFile, ivf.cc:1838-1916: faiss::read_index(filename))
            const std::string path = std::string("/tmp/knhip_test_") + c.name + ".faiss";
            FILE* f = std::fopen(path.c_str(), "wb");
            REQUIRE(f != nullptr);
            auto blob = bs.GetByName(c.name);
            REQUIRE(std::fwrite(blob->data.get(), 1, (size_t)blob->size, f) == (size_t)blob->size);
            std::fclose(f);
            auto idx3 = IndexFactory::Instance().Create<fp32>(c.name, version).value();
            REQUIRE(idx3.DeserializeFromFile(path) == Status::success);
            REQUIRE(idx3.Count() == idx.Count());
            auto r3 = idx3.Search(query_ds, c.cfg, nullptr);
            REQUIRE(r3.has_value());
            if (r3.has_value()) {
                int diff = 0;
                for (int64_t i = 0; i < nq; i++) diff += r3.value()->GetIds()[i] != results.value()->GetIds()[i];
                REQUIRE(diff == 0);
            }
            std::remove(path.c_str());
            auto idx4 = IndexFactory::Instance().Create<fp32>(c.name, version).value();
            REQUIRE(idx4.DeserializeFromFile("/tmp/knhip_no_such_file") == Status::disk_file_error);
        }
        // 7. range search: every index type (IVF_PQ with m != 32 through the plain ADC dump kernel; m = 32 again below)
        check_range(idx, c.cfg);
        // config validation
        Json bad = c.cfg;
        bad[meta::TOPK] = 100000;
        REQUIRE(idx.Search(query_ds, bad, nullptr).error() == Status::out_of_range_in_json);
        bad = c.cfg;
        bad[meta::TOPK] = 0;
        REQUIRE(idx.Search(query_ds, bad, nullptr).error() == Status::out_of_range_in_json);
        bad = c.cfg;
        bad[meta::METRIC_TYPE] = "HAMMING";
        {
            auto hidx = IndexFactory::Instance().Create<fp32>(c.name, version).value();
            REQUIRE(hidx.Build(train_ds, bad) == Status::invalid_metric_type);
            std::string msg;
            REQUIRE(IndexStaticFaced<fp32>::ConfigCheck(c.name, version, bad, msg) == Status::invalid_metric_type);
            REQUIRE(!msg.empty());
        }
        // 8. the static functions Milvus calls without an index instance (index_static.h:54-90)
        {
            std::string msg;
            REQUIRE(IndexStaticFaced<fp32>::ConfigCheck(c.name, version, c.cfg, msg) == Status::success);
            auto scfg = IndexStaticFaced<fp32>::CreateConfig(c.name, version);
            REQUIRE(scfg != nullptr);
            const bool raw = IndexStaticFaced<fp32>::HasRawData(c.name, version, c.cfg);
            const bool flat = std::string(c.name) == IndexEnum::INDEX_HIP_BRUTEFORCE ||
                              std::string(c.name) == IndexEnum::INDEX_HIP_IVFFLAT;
            REQUIRE(raw == flat);
            REQUIRE(idx.HasRawData(metric::L2) == flat);
        }
        // 9. GetVectorByIds on the kinds that keep the rows
        {
            const int64_t want[3] = {7, 4242, nb - 1};
            auto ids_ds = GenIdsDataSet(3, want);
            auto gv = idx.GetVectorByIds(ids_ds);
            if (idx.HasRawData(metric::L2)) {
                REQUIRE(gv.has_value());
                if (gv.has_value()) {
                    const float* got = static_cast<const float*>(gv.value()->GetTensor());
                    const float* base = static_cast<const float*>(train_ds->GetTensor());
                    int bad_rows = 0;
                    for (int r = 0; r < 3; r++) bad_rows += std::memcmp(got + r * dim, base + want[r] * dim, sizeof(float) * dim) != 0;
                    REQUIRE(bad_rows == 0);
                }
            } else {
                REQUIRE(!gv.has_value());
            }
        }
        // 10. cancellation (include/knowhere/context.h:24-29): a cancelled OpContext aborts the search
        {
            milvus::OpContext ctx;
            ctx.cancelled = true;
            REQUIRE(idx.Search(query_ds, c.cfg, nullptr, &ctx).error() == Status::timeout);
            milvus::OpContext live;
            REQUIRE(idx.Search(query_ds, c.cfg, nullptr, &live).has_value());
        }
        // 11. repeated Add (IndexNode::Add appends; ids continue at Count()): the second half is found under its ids
        {
            auto aidx = IndexFactory::Instance().Create<fp32>(c.name, version).value();
            const float* base = static_cast<const float*>(train_ds->GetTensor());
            auto first = knowhere::GenDataSet(nb / 2, dim, base);
            auto second = knowhere::GenDataSet(nb - nb / 2, dim, base + (nb / 2) * dim);
            first->SetIsOwner(false);
            second->SetIsOwner(false);
            REQUIRE(aidx.Train(train_ds, c.cfg) == Status::success);
            REQUIRE(aidx.Add(first, c.cfg) == Status::success);
            REQUIRE(aidx.Count() == nb / 2);
            REQUIRE(aidx.Add(second, c.cfg) == Status::success);
            REQUIRE(aidx.Count() == nb);
            auto r = aidx.Search(query_ds, c.cfg, nullptr);
            REQUIRE(r.has_value());
            int diff = 0;  // same training set, same rows: identical to the one-shot Build
            for (int64_t i = 0; i < nq; i++) diff += r.value()->GetIds()[i] != results.value()->GetIds()[i];
            REQUIRE(diff == 0);
        }
        // 12. id map (include/knowhere/index/index_node.h:292-318, id_map.h): the bitset is in public ids, results come
        //     back in public ids
        {
            auto midx = IndexFactory::Instance().Create<fp32>(c.name, version).value();
            REQUIRE(midx.Build(train_ds, c.cfg) == Status::success);
            std::vector<int64_t> in_to_out(nb);
            for (int64_t i = 0; i < nb; i++) in_to_out[i] = nb - 1 - i;  // storage id i is public id nb-1-i
            midx.Node()->GetIdMap().SetInToOut(in_to_out);
            auto r = midx.Search(query_ds, c.cfg, nullptr);
            REQUIRE(r.has_value());
            int diff = 0;
            for (int64_t i = 0; i < nq; i++) diff += r.value()->GetIds()[i] != nb - 1 - results.value()->GetIds()[i];
            REQUIRE(diff == 0);
            // filter the public ids of the first 40 % of storage rows' mirror: public id p filtered <=> p < 0.4 nb
            auto bits = BitsetFirst(nb, (size_t)(0.4 * nb));
            BitsetView bv(bits.data(), nb);
            auto rf = midx.Search(query_ds, c.cfg, bv);
            REQUIRE(rf.has_value());
            int leaked = 0;
            for (int64_t i = 0; i < nq; i++) leaked += rf.value()->GetIds()[i] >= 0 && bv.test(rf.value()->GetIds()[i]);
            REQUIRE(leaked == 0);
            // the same filter expressed in storage ids on the unmapped index gives the mirrored result
            std::vector<uint8_t> sbits((nb + 7) / 8, 0);
            for (int64_t i = 0; i < nb; i++) {
                if (bv.test(nb - 1 - i)) sbits[i >> 3] |= (1 << (i & 7));
            }
            auto rs = idx.Search(query_ds, c.cfg, BitsetView(sbits.data(), nb));
            REQUIRE(rs.has_value());
            diff = 0;
            for (int64_t i = 0; i < nq; i++) {
                const int64_t a = rf.value()->GetIds()[i], b = rs.value()->GetIds()[i];
                diff += a != (b < 0 ? b : nb - 1 - b);
            }
            REQUIRE(diff == 0);
        }
        // 13. the thread-pool wrapper bounds the searches in flight (index_node_thread_pool_wrapper.h; gpu_cuvs.h:48)
        {
            auto* wrapper = dynamic_cast<IndexNodeThreadPoolWrapper*>(idx.Node());
            REQUIRE(wrapper != nullptr);
            if (wrapper) {
                std::vector<std::thread> th;
                std::atomic<int> okc{0};
                for (int t = 0; t < 12; t++) {
                    th.emplace_back([&] {
                        for (int rep = 0; rep < 3; rep++) okc += idx.Search(query_ds, c.cfg, nullptr).has_value();
                    });
                }
                for (auto& t : th) t.join();
                REQUIRE(okc == 36);
                REQUIRE(wrapper->MaxInFlightSeen() >= 1 && wrapper->MaxInFlightSeen() <= wrapper->PoolSize());
                std::printf("   thread pool: size %zu, max in flight %zu\n", wrapper->PoolSize(), wrapper->MaxInFlightSeen());
            }
        }
        // 14. damaged blobs are rejected, not loaded (Deserialize cross-field validation)
        {
            auto blob = bs.GetByName(c.name);
            for (int64_t cut : {(int64_t)3, (int64_t)40, blob->size / 2, blob->size - 1}) {
                BinarySet tb;
                std::shared_ptr<uint8_t[]> copy(new uint8_t[cut]);
                std::memcpy(copy.get(), blob->data.get(), (size_t)cut);
                tb.Append(c.name, copy, cut);
                auto didx = IndexFactory::Instance().Create<fp32>(c.name, version).value();
                REQUIRE(didx.Deserialize(tb) != Status::success);
            }
            BinarySet empty;
            auto didx = IndexFactory::Instance().Create<fp32>(c.name, version).value();
            REQUIRE(didx.Deserialize(empty) == Status::invalid_binary_set);
        }
    }

    {   // IVF_PQ with m = 32: the range search path of the headline kernel
        Json cfg = ivfpq_gen();
        cfg[indexparam::M] = 32;
        auto idx = IndexFactory::Instance().Create<fp32>(IndexEnum::INDEX_HIP_IVFPQ, version).value();
        REQUIRE(idx.Build(train_ds, cfg) == Status::success);
        check_range(idx, cfg);
    }

    // COSINE (a12): self-search hits itself at ~1, results agree with the brute-force node, and the blob round-trips
    // (IxF9 for the flat index, IwFl + cosine inverted lists for IVF_FLAT: the stored norms travel)
    for (const char* name : {IndexEnum::INDEX_HIP_BRUTEFORCE, IndexEnum::INDEX_HIP_IVFFLAT, IndexEnum::INDEX_HIP_IVFPQ,
                             IndexEnum::INDEX_HIP_IVFSQ8}) {
        Json cfg = std::string(name) == IndexEnum::INDEX_HIP_IVFPQ ? ivfpq_gen() : ivfflat_gen();
        cfg[meta::METRIC_TYPE] = metric::COSINE;
        cfg[meta::TOPK] = 5;
        auto idx = IndexFactory::Instance().Create<fp32>(name, version).value();
        REQUIRE(idx.Build(train_ds, cfg) == Status::success);
        auto r = idx.Search(train_ds, cfg, nullptr);
        REQUIRE(r.has_value());
        if (!r.has_value()) continue;
        int bad = 0;
        for (int i = 0; i < nq; i++) bad += r.value()->GetIds()[i * 5] != i;
        REQUIRE(bad <= (std::string(name) == IndexEnum::INDEX_HIP_IVFPQ ? nq / 20 : 0));
        if (std::string(name) != IndexEnum::INDEX_HIP_IVFPQ && std::string(name) != IndexEnum::INDEX_HIP_IVFSQ8) {
            REQUIRE(std::abs(r.value()->GetDistance()[0] - 1.0f) < 1e-5f);
        }
        auto q = idx.Search(query_ds, cfg, nullptr);
        auto g = BruteForce::Search<fp32>(train_ds, query_ds, cfg, nullptr);
        REQUIRE(q.has_value() && g.has_value());
        float rc = GetKNNRecall(*g.value(), *q.value());
        std::printf("== %s COSINE recall@5 %.4f\n", name, rc);
        REQUIRE(rc >= (std::string(name) == IndexEnum::INDEX_HIP_IVFPQ ? 0.5f : 0.9f));
        BinarySet bs;
        REQUIRE(idx.Serialize(bs) == Status::success);
        if (std::string(name) == IndexEnum::INDEX_HIP_BRUTEFORCE) {
            REQUIRE(std::memcmp(bs.GetByName(name)->data.get(), "IxF9", 4) == 0);
        }
        auto idx2 = IndexFactory::Instance().Create<fp32>(name, version).value();
        REQUIRE(idx2.Deserialize(bs, cfg) == Status::success);
        auto q2 = idx2.Search(query_ds, cfg, nullptr);
        REQUIRE(q2.has_value());
        if (q2.has_value()) {
            int diff = 0;
            for (int64_t i = 0; i < nq * 5; i++) {
                diff += q2.value()->GetIds()[i] != q.value()->GetIds()[i];
                diff += q2.value()->GetDistance()[i] != q.value()->GetDistance()[i];
            }
            REQUIRE(diff == 0);
        }
    }

    {   // the static BruteForce functions beside Search (include/knowhere/comp/brute_force.h:41-58): SearchWithBuf fills the
        // caller's buffers with Search's answer; RangeSearch returns every row inside the radius and nothing else (checked
        // against the k = 64 search: a query's neighbours under the radius are exactly the prefix of its sorted list)
        Json cfg = base_gen();
        cfg[meta::TOPK] = 64;
        auto g = BruteForce::Search<fp32>(train_ds, query_ds, cfg, nullptr);
        REQUIRE(g.has_value());
        std::vector<int64_t> bi((size_t)nq * 64);
        std::vector<float> bd((size_t)nq * 64);
        REQUIRE(BruteForce::SearchWithBuf<fp32>(train_ds, query_ds, bi.data(), bd.data(), cfg, nullptr) == Status::success);
        int diff = 0;
        for (int64_t i = 0; i < nq * 64; i++) {
            diff += bi[i] != g.value()->GetIds()[i];
            diff += bd[i] != g.value()->GetDistance()[i];
        }
        REQUIRE(diff == 0);
        REQUIRE(BruteForce::SearchWithBuf<fp32>(train_ds, query_ds, nullptr, bd.data(), cfg, nullptr) == Status::invalid_args);
        std::vector<float> d8(nq);
        for (int64_t i = 0; i < nq; i++) d8[i] = g.value()->GetDistance()[i * 64 + 8];
        std::nth_element(d8.begin(), d8.begin() + nq / 2, d8.end());
        const float radius = d8[nq / 2];  // (about 8 neighbours per query: far inside the 64 of the search)
        Json rcfg = base_gen();
        rcfg[meta::RADIUS] = radius;
        rcfg[meta::RANGE_FILTER] = 0.0f;
        auto rr = BruteForce::RangeSearch<fp32>(train_ds, query_ds, rcfg, nullptr);
        REQUIRE(rr.has_value());
        if (rr.has_value()) {
            const size_t* lims = rr.value()->GetLims();
            int wrong = 0;
            for (int64_t i = 0; i < nq; i++) {
                int64_t want = 0;
                while (want < 64 && g.value()->GetDistance()[i * 64 + want] < radius) want++;
                if (want == 64) continue;  // (more than the search saw: not decidable from it)
                wrong += (int64_t)(lims[i + 1] - lims[i]) != want;
                for (size_t j = lims[i]; j < lims[i + 1]; j++) {
                    bool found = false;
                    for (int64_t t = 0; t < want; t++) found |= g.value()->GetIds()[i * 64 + t] == rr.value()->GetIds()[j];
                    wrong += !found;
                }
            }
            REQUIRE(wrong == 0);
            std::printf("== BruteForce::RangeSearch: %zu results inside radius %.3f\n", lims[nq], radius);
        }
    }

    {   // config limits of the backend (hip_index_node.h): m, dim % m, nbits outside 1 .. 8, sq_type
        std::string msg;
        Json cfg = ivfpq_gen();
        cfg[indexparam::M] = 12;
        REQUIRE(IndexStaticFaced<fp32>::ConfigCheck(IndexEnum::INDEX_HIP_IVFPQ, version, cfg, msg) == Status::invalid_args);
        cfg[indexparam::M] = 64;
        cfg[meta::DIM] = 96;
        REQUIRE(IndexStaticFaced<fp32>::ConfigCheck(IndexEnum::INDEX_HIP_IVFPQ, version, cfg, msg) == Status::invalid_args);
        cfg = ivfpq_gen();
        cfg[indexparam::NBITS] = 4;  // codes of 1 .. 8 bits since round 6 (the cuVS config accepts 4 .. 8)
        REQUIRE(IndexStaticFaced<fp32>::ConfigCheck(IndexEnum::INDEX_HIP_IVFPQ, version, cfg, msg) == Status::success);
        cfg[indexparam::NBITS] = 12;  // wider than a byte: refused by the config range
        REQUIRE(IndexStaticFaced<fp32>::ConfigCheck(IndexEnum::INDEX_HIP_IVFPQ, version, cfg, msg) ==
                Status::out_of_range_in_json);
        cfg = ivfflat_gen();
        cfg[indexparam::SQ_TYPE] = "FP16";
        REQUIRE(IndexStaticFaced<fp32>::ConfigCheck(IndexEnum::INDEX_HIP_IVFSQ8, version, cfg, msg) == Status::invalid_args);
        cfg[indexparam::SQ_TYPE] = "sq8";
        REQUIRE(IndexStaticFaced<fp32>::ConfigCheck(IndexEnum::INDEX_HIP_IVFSQ8, version, cfg, msg) == Status::success);
        REQUIRE(IndexStaticFaced<fp32>::ConfigCheck("NO_SUCH_INDEX", version, cfg, msg) == Status::invalid_index_error);
        // string-typed numbers are accepted as Milvus sends them (Config::FormatAndCheck)
        cfg = ivfflat_gen();
        cfg[indexparam::NLIST] = "20";
        cfg[meta::DIM] = "128";
        REQUIRE(IndexStaticFaced<fp32>::ConfigCheck(IndexEnum::INDEX_HIP_IVFFLAT, version, cfg, msg) == Status::success);
        cfg[indexparam::NLIST] = "twenty";
        REQUIRE(IndexStaticFaced<fp32>::ConfigCheck(IndexEnum::INDEX_HIP_IVFFLAT, version, cfg, msg) ==
                Status::invalid_value_in_json);
    }

    std::printf("%s: %d checks, %d failed\n", g_fail ? "FAILED" : "PASSED", g_checks, g_fail);
    return g_fail ? 1 : 0;
}
