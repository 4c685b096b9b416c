// pairec_host.cpp — implementation of the C++ host mirror (see pairec_host.hpp) and the small C
// driver API (ph_*) that the Python tests use to exercise it.
#include "pairec_host.hpp"

#include <algorithm>
#include <chrono>
#include <cerrno>
#include <cmath>
#include <cstring>
#include <ctime>
#include <thread>

namespace pairec {

// ---- utils ---------------------------------------------------------------------------------------
double ToFloat(const json::Value& v, double def) {          // utils/type.go:43-69
    if (v.type == json::Value::Number) return v.num;
    if (v.type == json::Value::String) {
        char* e = nullptr;
        const double d = strtod(v.str.c_str(), &e);
        if (e != v.str.c_str() && *e == '\0') return d;
        return def;
    }
    return def;
}

namespace module {
bool Item::FloatExprData(const std::string& name, double* out) {
    if (name == "current_score") {                          // item.go:190-197
        algoScores["recall_score"] = Score;
        *out = Score;
        return true;
    }
    auto a = algoScores.find(name);
    if (a != algoScores.end()) { *out = a->second; return true; }
    auto p = Properties.find(name);
    if (p != Properties.end()) { *out = ToFloat(p->second, 0.0); return true; }
    return false;
}
bool InMemoryVectorDao::VectorString(const std::string& id, std::string* out, std::string* err) {
    auto it = vectors.find(id);
    if (it == vectors.end() || it->second.empty()) {
        if (err) *err = "vector empty";                     // module.VectoryEmptyError
        return false;
    }
    *out = it->second;
    return true;
}
}  // namespace module

namespace context {
std::string RecommendContext::GetParameter(const std::string& k) const {
    auto it = Param.find(k);
    return (it != Param.end() && it->second.type == json::Value::String) ? it->second.str : "";
}
double RecommendContext::GetFloat(const std::string& k, double def) const {
    if (HasExperiment() && ExperimentParamsJson.has(k)) return ToFloat(ExperimentParamsJson.at(k), def);
    auto it = ExperimentParams.find(k);
    return it != ExperimentParams.end() ? it->second : def;
}
long long RecommendContext::GetInt(const std::string& k, long long def) const {
    if (HasExperiment() && ExperimentParamsJson.has(k)) return (long long)ToFloat(ExperimentParamsJson.at(k), (double)def);
    auto it = ExperimentParams.find(k);
    return it != ExperimentParams.end() ? (long long)it->second : def;
}
}  // namespace context

// ---- fmt %v for float64 ------------------------------------------------------------------------------
std::string GoFmtFloat(double x) {
    if (x != x) return "NaN";
    if (std::isinf(x)) return x > 0 ? "+Inf" : "-Inf";
    if (x == 0) return std::signbit(x) ? "-0" : "0";
    // shortest digit string that round-trips (strconv 'g', precision -1)
    char buf[40];
    int prec = 1;
    for (; prec <= 17; ++prec) {
        snprintf(buf, sizeof buf, "%.*e", prec - 1, x);
        if (strtod(buf, nullptr) == x) break;
    }
    // buf = [-]d.ddddde[+-]XX
    std::string m(buf);
    const size_t epos = m.find('e');
    const int dexp = atoi(m.c_str() + epos + 1);
    std::string digits;
    for (size_t i = 0; i < epos; ++i)
        if (m[i] >= '0' && m[i] <= '9') digits.push_back(m[i]);
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    const std::string sign = x < 0 ? "-" : "";
    if (dexp < -4 || dexp >= 6) {               // strconv 'g', shortest: the %e form from exponent 6 on
        std::string out = sign + digits.substr(0, 1);
        if (digits.size() > 1) out += "." + digits.substr(1);
        char e[16];
        snprintf(e, sizeof e, "e%c%02d", dexp >= 0 ? '+' : '-', std::abs(dexp));
        return out + e;
    }
    if (dexp >= 0) {
        if ((int)digits.size() <= dexp + 1) return sign + digits + std::string((size_t)(dexp + 1) - digits.size(), '0');
        return sign + digits.substr(0, (size_t)dexp + 1) + "." + digits.substr((size_t)dexp + 1);
    }
    return sign + "0." + std::string((size_t)(-dexp - 1), '0') + digits;
}

// ---- cache --------------------------------------------------------------------------------------------
namespace cache {
namespace {
struct MemCache : Cache {            // go-cache semantics: per-entry expiry, lazy eviction on Get
    Value::Kind kind;
    struct Entry { std::string val; std::chrono::steady_clock::time_point until; bool forever; };
    std::mutex mu;
    std::map<std::string, Entry> m;
    explicit MemCache(Value::Kind k) : kind(k) {}
    void Put(const std::string& key, const std::string& val, int ttl) override {
        std::lock_guard<std::mutex> g(mu);
        m[key] = Entry{val, std::chrono::steady_clock::now() + std::chrono::seconds(ttl), ttl <= 0};
    }
    Value Get(const std::string& key) override {
        std::lock_guard<std::mutex> g(mu);
        auto it = m.find(key);
        if (it == m.end()) return Value{};
        if (!it->second.forever && std::chrono::steady_clock::now() >= it->second.until) {
            m.erase(it);
            return Value{};
        }
        return Value{kind, it->second.val};
    }
};
}  // namespace
std::shared_ptr<Cache> NewCache(const std::string& adapter, const std::string&, std::string* err) {
    if (adapter == "localCache") return std::make_shared<MemCache>(Value::kString);
    if (adapter == "localBytes") return std::make_shared<MemCache>(Value::kBytes);
    if (err) *err = "Cache:not found instance, name:" + adapter;           // cache.go:30
    return nullptr;
}
}  // namespace cache

// ---- recconf -------------------------------------------------------------------------------------
namespace recconf {
static std::vector<std::string> str_list(const json::Value& v) {
    std::vector<std::string> out;
    if (v.type == json::Value::Array)
        for (const auto& e : v.arr)
            if (e.type == json::Value::String) out.push_back(e.str);
    return out;
}
bool RecommendConfig::Parse(const std::string& text, RecommendConfig* out, std::string* err) {
    json::Value root;
    if (!json::Parser(text).Parse(&root, err)) return false;
    if (root.type != json::Value::Object) { if (err) *err = "recconf: root is not an object"; return false; }
    for (const auto& a : root.at("AlgoConfs").arr) out->AlgoConfs.push_back({a.s("Name"), a.s("Type"), a});
    for (const auto& r : root.at("RecallConfs").arr) {
        RecallConfig c;
        c.Name = r.s("Name"); c.RecallType = r.s("RecallType"); c.RecallAlgo = r.s("RecallAlgo");
        c.ItemType = r.s("ItemType"); c.CachePrefix = r.s("CachePrefix");
        c.CacheAdapter = r.s("CacheAdapter"); c.CacheConfig = r.s("CacheConfig");
        c.RecallCount = (int)r.n("RecallCount"); c.CacheTime = (int)r.n("CacheTime");
        c.DaoAdapterType = r.at("DaoConf").s("AdapterType");
        c.VectorDaoAdapterType = r.at("VectorDaoConf").s("AdapterType");
        c.HologresName = r.at("VectorDaoConf").s("HologresName");
        c.VectorAlgoType = r.s("VectorAlgoType");
        out->RecallConfs.push_back(c);
    }
    for (const auto& kv : root.at("RankConf").obj) {
        RankConfig c;
        c.RankAlgoList = str_list(kv.second.at("RankAlgoList"));
        c.RankScore = kv.second.s("RankScore"); c.Processor = kv.second.s("Processor");
        c.ASTType = kv.second.s("ASTType"); c.BatchCount = (int)kv.second.n("BatchCount");
        for (const auto& rw : kv.second.at("ScoreRewrite").obj) c.ScoreRewrite[rw.first] = rw.second.str;
        out->RankConf[kv.first] = c;
    }
    for (const auto& kv : root.at("SortNames").obj) out->SortNames[kv.first] = str_list(kv.second);
    auto parse_scene_features = [&](const json::Value& scenes, std::map<std::string, SceneFeatureConfig>* dst) {
        for (const auto& sc : scenes.obj) {                        // recconf.go:52-53,169-176,256-265
            SceneFeatureConfig sf;
            sf.AsynLoadFeature = sc.second.at("AsynLoadFeature").type == json::Value::Bool && sc.second.at("AsynLoadFeature").b;
            for (const auto& lc : sc.second.at("FeatureLoadConfs").arr) {
                FeatureLoadConfig l;
                l.DaoAdapterType = lc.at("FeatureDaoConf").s("AdapterType");
                for (const auto& f : lc.at("Features").arr) {
                    FeatureConfig c;
                    c.FeatureType = f.s("FeatureType"); c.FeatureName = f.s("FeatureName"); c.FeatureSource = f.s("FeatureSource");
                    c.FeatureValue = f.s("FeatureValue"); c.FeatureStore = f.s("FeatureStore"); c.Normalizer = f.s("Normalizer");
                    c.Expression = f.s("Expression");
                    c.RemoveFeatureSource = f.at("RemoveFeatureSource").type == json::Value::Bool && f.at("RemoveFeatureSource").b;
                    l.Features.push_back(std::move(c));
                }
                sf.FeatureLoadConfs.push_back(std::move(l));
            }
            (*dst)[sc.first] = std::move(sf);
        }
    };
    parse_scene_features(root.at("FeatureConfs"), &out->FeatureConfs);
    parse_scene_features(root.at("UserFeatureConfs"), &out->UserFeatureConfs);
    for (const auto& sc : root.at("SceneConfs").obj)
        for (const auto& cat : sc.second.obj)
            out->SceneRecallNames[sc.first][cat.first] = str_list(cat.second.at("RecallNames"));
    auto is_false = [](std::string v) {
        std::transform(v.begin(), v.end(), v.begin(), ::tolower);
        return v == "false";
    };
    auto parse_dpp = [&](const json::Value& d, const std::string& name) {
        DPPSortConfig c;
        c.Name = name; c.Alpha = d.d("Alpha", 1.0); c.WindowSize = (int)d.n("WindowSize");
        c.CandidateCount = (int)d.n("CandidateCount"); c.MinScorePercent = d.d("MinScorePercent");
        c.NormalizeEmb = !is_false(d.s("NormalizeEmb"));    // dpp_sort.go:95-97
        c.EnsurePositiveSim = !is_false(d.s("EnsurePositiveSim"));   // :98-100
        c.AbortRunCount = (int)d.n("AbortRunCount");
        c.FilterRetrieveIds = str_list(d.at("FilterRetrieveIds"));
        c.EmbeddingHookNames = str_list(d.at("EmbeddingHookNames"));
        return c;
    };
    for (const auto& d : root.at("DPPConf").arr) out->DPPConf.push_back(parse_dpp(d, d.s("Name")));
    // SortConfs (recconf.go:86, sort/sort.go:162-200); the GPU diversity sorts of a pairec process are declared in
    // UserDefineConfs.pairec_gpu.Sorts with the same nested DPPConf / SSDConf objects (Engine::Create)
    auto parse_sort = [&](const json::Value& sc) {
        SortConfig c;
        c.Name = sc.s("Name"); c.SortType = sc.s("SortType");
        c.SortByField = sc.s("SortByField");
        c.SortOrder = sc.s("SortOrder");
        c.SwitchThreshold = sc.d("SwitchThreshold");
        if (c.SortType == "DPPSort") {
            c.DPPConf = parse_dpp(sc.at("DPPConf"), c.Name);
            c.HologresName = sc.at("DPPConf").at("DaoConf").s("HologresName");
        } else if (c.SortType == "SSDSort") {
            const json::Value& d = sc.at("SSDConf");
            c.HologresName = d.at("DaoConf").s("HologresName");
            SSDSortConfig& s = c.SSDConf;
            s.Name = c.Name;
            if (d.d("Gamma") > 0) s.Gamma = d.d("Gamma");                         // ssd_sort.go:81-83
            s.UseSSDStar = d.at("UseSSDStar").type == json::Value::Bool && d.at("UseSSDStar").b;
            s.WindowSize = (int)d.n("WindowSize") > 0 ? (int)d.n("WindowSize") : 5; // :84-86
            s.AbortRunCount = (int)d.n("AbortRunCount");
            s.CandidateCount = (int)d.n("CandidateCount");
            s.MinScorePercent = d.d("MinScorePercent");
            s.NormalizeEmb = !is_false(d.s("NormalizeEmb"));                      // :90-92
            s.EnsurePositiveSim = !is_false(d.s("EnsurePositiveSim"));            // :93-95
            s.FilterRetrieveIds = str_list(d.at("FilterRetrieveIds"));
        }
        return c;
    };
    for (const auto& sc : root.at("SortConfs").arr) out->SortConfs.push_back(parse_sort(sc));
    out->UserDefineConfs = root.at("UserDefineConfs");
    // the GPU plug-ins' own declarations, in the shapes of RecallConfig / SortConfig
    for (const auto& r : out->UserDefineConfs.at("pairec_gpu").at("Recalls").arr) {
        RecallConfig c;
        c.Name = r.s("Name"); c.Kind = r.s("Kind", "vector"); c.RecallAlgo = r.s("RecallAlgo");
        c.ItemType = r.s("ItemType"); c.CachePrefix = r.s("CachePrefix");
        c.CacheAdapter = r.s("CacheAdapter"); c.CacheConfig = r.s("CacheConfig");
        c.RecallCount = (int)r.n("RecallCount"); c.CacheTime = (int)r.n("CacheTime");
        c.RankScore = r.s("RankScore"); c.RankVar = r.s("RankVar");
        for (const auto& rw : r.at("ScoreRewrite").obj) c.ScoreRewrite[rw.first] = rw.second.str;
        // HologresVectorConf.WhereClause / TimeInterval (recconf.go:492-497) restrict the SQL's candidates.  The device serves
        // the form `column OP integer` over an int32 item column keyed by row, "${time}" standing for now - TimeInterval as in
        // hologres_vector_recall.go:56-61; anything else is refused here rather than answered over other candidates.
        c.WhereClause = !r.s("WhereClause").empty() ? r.s("WhereClause") : r.at("HologresVectorConf").s("WhereClause");
        c.TimeInterval = (int)(r.n("TimeInterval") != 0 ? r.n("TimeInterval") : r.at("HologresVectorConf").n("TimeInterval"));
        if (!c.WhereClause.empty()) {
            if (c.Kind != "hologres" && c.Kind != "hologres_v2" && c.Kind != "online_hologres") {
                if (err) *err = "pairec_gpu.Recalls: " + c.Name + ": WhereClause is not supported by Kind \"" + c.Kind + "\" (Kinds hologres, hologres_v2, online_hologres take one)";
                return false;
            }
            if (!recall::ParseWhereClause(c.WhereClause, c.TimeInterval, &c.WhereColumn, &c.WhereOp, &c.WhereValue)) {
                if (err) *err = "pairec_gpu.Recalls: " + c.Name + ": WhereClause \"" + c.WhereClause +
                                "\" is not supported (the device serves `column OP integer`, OP one of > >= < <= = != <>, the integer or ${time})";
                return false;
            }
        }
        out->GpuRecalls.push_back(c);
    }
    for (const auto& sc : out->UserDefineConfs.at("pairec_gpu").at("Sorts").arr) out->GpuSorts.push_back(parse_sort(sc));
    return true;
}
}  // namespace recconf

// ---- algorithm -----------------------------------------------------------------------------------
namespace algorithm {
void AlgorithmFactory::RegisterAlgorithm(const std::string& name, std::shared_ptr<IAlgorithm> a) {
    std::lock_guard<std::mutex> g(mu_);
    algos_[name] = std::move(a);                            // overwrites (algorithm.go:164-168)
}
bool AlgorithmFactory::Run(const std::string& name, const AlgoData& data, AlgoResult* out, std::string* err) {
    std::shared_ptr<IAlgorithm> a;
    {
        std::lock_guard<std::mutex> g(mu_);
        auto it = algos_.find(name);
        if (it == algos_.end()) {
            if (err) *err = "not find algorithm, name:" + name;     // algorithm.go:113-115
            return false;
        }
        a = it->second;
    }
    return a->Run(data, out, err);
}
}  // namespace algorithm

// ---- response decoders ---------------------------------------------------------------------------
namespace algorithm {
namespace decode {
std::vector<AlgoResponse> EasyrecResponse(const std::vector<std::string>& item_ids,
                                          const std::map<std::string, std::vector<double>>& results) {
    std::vector<AlgoResponse> ret;                          // easyrec_response.go:226-232
    for (const auto& id : item_ids) {
        auto it = results.find(id);
        ret.emplace_back(it != results.end() && !it->second.empty() ? it->second[0] : 0.0);
    }
    return ret;
}
bool EasyrecMutValResponse(const std::vector<std::string>& item_ids, const std::vector<std::string>& outputs,
                           const std::map<std::string, std::vector<double>>& results,
                           std::vector<AlgoResponse>* out, std::string* err) {
    out->clear();                                           // easyrec_response.go:41-69
    for (const auto& id : item_ids) {
        AlgoResponse r;
        r.multiValModule = true;
        auto it = results.find(id);
        if (it != results.end()) {
            if (outputs.size() != it->second.size()) {
                if (err) *err = "outputs size is not equal scores";
                out->clear();
                return false;
            }
            for (size_t k = 0; k < outputs.size(); ++k) r.scoreArr[outputs[k]] = it->second[k];
        } else {
            for (const auto& o : outputs) r.scoreArr[o] = 0.0;
        }
        out->push_back(std::move(r));
    }
    return true;
}
bool EasyrecMutClassificationResponse(const std::vector<std::string>& item_ids,
                                      const std::map<std::string, std::pair<std::vector<float>, std::vector<long long>>>& tf_outputs,
                                      std::vector<AlgoResponse>* out, std::string* err) {
    out->assign(item_ids.size(), AlgoResponse());           // easyrec_response.go:145-199
    for (const auto& kv : tf_outputs) {
        const auto& vals = kv.second.first;
        const auto& shape = kv.second.second;
        const size_t width = shape.size() >= 2 ? (size_t)shape[1] : 1;        // [N, C] → C per item; [N] → 1
        if (vals.size() < item_ids.size() * width) {
            if (err) *err = "output " + kv.first + " holds fewer values than items";
            out->clear();
            return false;
        }
        for (size_t i = 0; i < item_ids.size(); ++i) {
            std::vector<double>& dst = (*out)[i].mulClassifyArr[kv.first];
            for (size_t c = 0; c < width; ++c) dst.push_back((double)vals[i * width + c]);   // float32 → float64
        }
    }
    return true;
}
double AlinkFMScore(double prediction_result, double prediction_score) {
    return prediction_result == 0.0 ? 1 - prediction_score : prediction_score;     // fm_response.go:28-34
}
std::vector<AlgoResponse> TFServingResponse(const std::vector<std::vector<double>>& outputs) {
    std::vector<AlgoResponse> ret;                          // tfserving/response.go:58-62
    for (const auto& val : outputs)
        for (double score : val) ret.emplace_back(score);
    return ret;
}
std::vector<AlgoResponse> WidenF32(const float* scores, size_t n) {
    std::vector<AlgoResponse> ret;
    for (size_t i = 0; i < n; ++i) ret.emplace_back((double)scores[i]);
    return ret;
}
std::vector<AlgoResponse> TfResponse(const std::vector<std::pair<std::string, OutputArray>>& outputs) {
    std::vector<AlgoResponse> ret;                          // tf_response.go:56-61: the first output only (`break`)
    if (!outputs.empty())
        for (float v : outputs[0].second.float_val) ret.emplace_back((double)v);
    return ret;
}
std::vector<AlgoResponse> TfMutValResponse(const std::vector<std::pair<std::string, OutputArray>>& outputs) {
    std::vector<AlgoResponse> ret;                          // tf_response.go:35-47
    for (const auto& kv : outputs)
        for (size_t i = 0; i < kv.second.float_val.size(); ++i) {
            if (i >= ret.size()) { ret.emplace_back(); ret.back().multiValModule = true; }
            ret[i].scoreArr[kv.first] = (double)kv.second.float_val[i];
        }
    return ret;
}
bool TorchrecMutValResponse(size_t n_items, const std::vector<std::pair<std::string, OutputArray>>& outputs,
                            std::vector<AlgoResponse>* out, std::string* err) {
    out->assign(n_items, AlgoResponse());                   // easyrec_response.go:474-491
    for (auto& r : *out) r.multiValModule = true;
    for (const auto& kv : outputs) {
        if (kv.second.size() < n_items) {                   // (the reference indexes FloatVal[i] and panics)
            if (err) *err = "output " + kv.first + " holds fewer values than items";
            out->clear();
            return false;
        }
        for (size_t i = 0; i < n_items; ++i) (*out)[i].scoreArr[kv.first] = kv.second.at(i);
    }
    return true;
}
bool TorchrecMutClassificationResponse(size_t n_items, const std::vector<std::pair<std::string, OutputArray>>& outputs,
                                       std::vector<AlgoResponse>* out, std::string* err) {
    out->assign(n_items, AlgoResponse());                   // easyrec_response.go:543-565
    for (const auto& kv : outputs) {
        const size_t dims = kv.second.shape.size();
        if (dims != 1 && dims != 2) continue;               // other ranks leave the output out of the map
        const size_t width = dims == 2 ? (size_t)kv.second.shape[1] : 1;
        if (kv.second.float_val.size() < n_items * width) {
            if (err) *err = "output " + kv.first + " holds fewer values than items";
            out->clear();
            return false;
        }
        for (size_t i = 0; i < n_items; ++i) {
            std::vector<double>& dst = (*out)[i].mulClassifyArr[kv.first];
            for (size_t c = 0; c < width; ++c) dst.push_back((double)kv.second.float_val[i * width + c]);
        }
    }
    return true;
}
bool TorchrecEmbeddingItemsResponse(const std::vector<std::string>& item_ids, const OutputArray* scores,
                                    std::vector<EmbeddingInfo>* out, std::string* err) {
    out->clear();                                           // easyrec_response.go:707-731
    for (size_t i = 0; i < item_ids.size(); ++i) {
        EmbeddingInfo info;
        info.ItemId = item_ids[i];
        if (scores) {
            if (i >= scores->size()) {
                if (err) *err = "match_item_scores holds fewer values than item_ids";
                out->clear();
                return false;
            }
            info.Score = scores->at(i);
        }
        out->push_back(info);
    }
    return true;
}
double PssmartScore(double score, const std::string& lable, const std::string& label) {
    return (lable == "0" || label == "0") ? 1 - score : score;       // pmml_response.go:24-31
}
}  // namespace decode
}  // namespace algorithm

// ---- recall --------------------------------------------------------------------------------------
namespace recall {
LoadOutcome CheckRecallConf(const recconf::RecallConfig& c) {
    const std::string& t = c.RecallType;
    auto panic = [](const std::string& m) { return LoadOutcome{LoadOutcome::kPanic, m}; };
    auto unavailable = [&](const std::string& what) {
        return LoadOutcome{LoadOutcome::kUnavailable, "recall " + c.Name + ": " + what + " is not available in the standalone mirror (no datasources)"};
    };
    auto one_of = [](const std::string& v, std::initializer_list<const char*> l) {
        for (const char* x : l) if (v == x) return true;
        return false;
    };
    if (t == "MockRecall" || t == "OnlineVectorRecall") return LoadOutcome{};       // constructors touch no datasource
    if (t == "VectorRecall") {                                // module.NewVectorDao (module/vector_dao.go:17-33)
        if (one_of(c.DaoAdapterType, {"redis", "hbase"}) || one_of(c.VectorDaoAdapterType, {"hologres", "mysql", "clickhouse", "be"}))
            return unavailable("a VectorDao over " + (c.DaoAdapterType.empty() ? c.VectorDaoAdapterType : c.DaoAdapterType));
        return panic("not found VectorDao implement");
    }
    if (t == "UserCustomRecall") {                            // module.NewUserCustomRecallDao (user_custom_recall_dao.go:12-28)
        if (one_of(c.DaoAdapterType, {"mysql", "tablestore", "hologres", "redis", "clickhouse", "featurestore"}))
            return unavailable("a UserCustomRecallDao over " + c.DaoAdapterType);
        return panic("not found UserCustomRecallDao implement");
    }
    if (one_of(t, {"HologresVectorRecall", "HologresVectorRecallV2", "I2IVectorRecall", "OnlineHologresVectorRecall"}))
        // holo.GetPostgres(VectorDaoConf.HologresName) → panic(err) (persist/holo: "Postgres not found, name:…")
        return panic("Postgres not found, name:" + c.HologresName);
    if (one_of(t, {"UserCollaborativeFilterRecall", "UserTopicRecall", "ItemCollaborativeFilterRecall", "UserGroupHotRecall",
                   "UserGlobalHotRecall", "ColdStartRecall", "BeRecall", "RealTimeU2IRecall", "GraphRecall", "OpenSearchRecall",
                   "RecallEngineRecall"}))
        return c.DaoAdapterType.empty() ? panic("not found " + t + " DAO implement (the constructor's DAO factory panics without DaoConf.AdapterType)")
                                        : unavailable("a DAO over " + c.DaoAdapterType);
    // unknown type — and "MilvusVectorRecall", whose constructor call is commented out (recall.go:78-79): recall stays nil
    return panic("recall empty, name:" + c.Name);
}
std::shared_ptr<Recall> Registry::GetRecall(const std::string& name, std::string* err) {
    auto it = recalls_.find(name);
    if (it == recalls_.end()) {
        if (err) *err = "recall:not found, name:" + name;
        return nullptr;
    }
    return it->second;
}
bool ParseWhereClause(const std::string& s, int time_interval, std::string* column, int* op, long long* value) {
    size_t p = 0;
    auto skip = [&]() { while (p < s.size() && isspace((unsigned char)s[p])) ++p; };
    skip();
    const size_t c0 = p;
    while (p < s.size() && (isalnum((unsigned char)s[p]) || s[p] == '_')) ++p;
    if (p == c0 || isdigit((unsigned char)s[c0])) return false;
    *column = s.substr(c0, p - c0);
    skip();
    static const struct { const char* text; int op; } kOps[] = {{">=", 1}, {"<=", 3}, {"!=", 5}, {"<>", 5}, {"==", 4}, {">", 0}, {"<", 2}, {"=", 4}};
    *op = -1;
    for (const auto& o : kOps)
        if (s.compare(p, strlen(o.text), o.text) == 0) { *op = o.op; p += strlen(o.text); break; }
    if (*op < 0) return false;
    skip();
    if (s.compare(p, 7, "${time}") == 0) {
        *value = (long long)time(nullptr) - (long long)time_interval;
        p += 7;
    } else {
        char* end = nullptr;
        errno = 0;
        *value = strtoll(s.c_str() + p, &end, 10);
        if (end == s.c_str() + p || errno) return false;
        p = (size_t)(end - s.c_str());
    }
    skip();
    return p == s.size();
}

std::vector<float> ParseVectorString(const std::string& s) {     // vector_recall.go:70-82
    std::vector<float> out;
    size_t p = 0;
    while (p <= s.size()) {
        size_t e = s.find(' ', p);
        if (e == std::string::npos) e = s.size();
        const std::string vc = s.substr(p, e - p);
        p = e + 1;
        const size_t c = vc.find(':');
        if (c == std::string::npos) continue;
        if (vc.find(':', c + 1) != std::string::npos) continue;       // len(vals) != 2
        const std::string val = vc.substr(c + 1);
        char* end = nullptr;
        const double d = strtod(val.c_str(), &end);
        // `value, _ := strconv.ParseFloat(vals[1], 32)`: a malformed number yields 0
        out.push_back((end != val.c_str() && *end == '\0') ? (float)d : 0.0f);
    }
    return out;
}
std::string FormatCacheString(const std::vector<module::ItemPtr>& items, const std::string& recall_name) {
    std::string out;                                               // vector_recall.go:105-110
    for (size_t i = 0; i < items.size(); ++i) {
        if (i) out.push_back(',');
        out += items[i]->Id + ":" + recall_name + ":" + GoFmtFloat(items[i]->Score);
    }
    return out;
}
bool ParseCacheString(const std::string& line, const std::string& recall_name, const std::string& item_type,
                      std::vector<module::ItemPtr>* out, std::string* err) {
    out->clear();                                                  // vector_recall.go:39-55
    size_t p = 0;
    while (p <= line.size()) {
        size_t e = line.find(',', p);
        if (e == std::string::npos) e = line.size();
        const std::string id = line.substr(p, e - p);
        p = e + 1;
        module::ItemPtr item;
        if (id.find(':') != std::string::npos) {
            std::vector<std::string> vars;
            size_t q = 0;
            while (q <= id.size()) {
                size_t c = id.find(':', q);
                if (c == std::string::npos) c = id.size();
                vars.push_back(id.substr(q, c - q));
                q = c + 1;
            }
            if (vars.size() < 3) {      // `vars[2]` would panic the reference (index out of range)
                if (err) *err = "recall cache entry \"" + id + "\": expected id:name:score";
                out->clear();
                return false;
            }
            item = std::make_shared<module::Item>(vars[0]);
            char* end = nullptr;
            const double f = strtod(vars[2].c_str(), &end);        // `f, _ := strconv.ParseFloat(vars[2], 64)`
            item->Score = (end != vars[2].c_str() && *end == '\0') ? f : 0.0;
        } else {
            item = std::make_shared<module::Item>(id);
        }
        item->RetrieveId = recall_name;
        item->ItemType = item_type;
        out->push_back(item);
        if (e == line.size()) break;
    }
    return true;
}
}  // namespace recall

// ---- filter --------------------------------------------------------------------------------------
namespace filter {
std::vector<module::ItemPtr> UniqueFilter(const std::vector<module::ItemPtr>& items) {
    std::vector<module::ItemPtr> out;
    std::map<module::ItemId, module::ItemPtr> uniq;
    for (const auto& item : items) {
        auto it = uniq.find(item->Id);
        if (it == uniq.end()) {
            uniq[item->Id] = item;
            out.push_back(item);
        } else {
            auto& exist = it->second;
            for (const auto& kv : item->algoScores) exist->AddAlgoScore(kv.first, kv.second);
            if (!exist->hasRecallScores) {
                exist->RecallScores = {{exist->RetrieveId, exist->Score}};
                exist->hasRecallScores = true;
            }
            exist->RecallScores[item->RetrieveId] = item->Score;
        }
    }
    return out;
}
}  // namespace filter

// ---- sort ----------------------------------------------------------------------------------------
namespace sort {
bool Registry::RegisterSort(const std::string& name, std::shared_ptr<ISort> s, std::string* err) {
    if (!s) { if (err) *err = "Sort is nil, name:" + name; return false; }      // sort.go:144-146 panics
    if (sorts_.count(name) == 0) sorts_[name] = std::move(s);                   // first registration wins
    return true;
}
std::shared_ptr<ISort> Registry::Get(const std::string& name) {
    auto it = sorts_.find(name);
    return it == sorts_.end() ? nullptr : it->second;
}
}  // namespace sort

// ---- GPU-backed plugins --------------------------------------------------------------------------
namespace {

std::string pg_err(const char* what) { return std::string(what) + ": " + pg_last_error(); }

// algorithm.IAlgorithm replacing FaissModel (algorithm/faiss/model.go:29-31)
struct GpuFaissAlgorithm : algorithm::IAlgorithm {
    Engine* e;
    explicit GpuFaissAlgorithm(Engine* eng) : e(eng) {}
    bool Init(const recconf::AlgoConfig&, std::string*) override { return true; }
    bool Run(const algorithm::AlgoData& data, algorithm::AlgoResult* out, std::string* err) override {
        if (data.kind != algorithm::AlgoData::kVector) { if (err) *err = "faiss: invalid request type"; return false; }
        const auto& req = data.vec;
        if (req.Vector.size() != e->dim) { if (err) *err = "faiss: vector dimension mismatch"; return false; }
        if (req.K == 0) return true;
        std::vector<uint64_t> rows(req.K);
        std::vector<float> scores(req.K);
        uint32_t cnt = 0;
        if (e->coalesce) {
            // one request per call, many calls at once (one goroutine per recall, service/recall.go:129-145): the library
            // batches them into shared table passes
            pg_coalescer* co = e->SceneCoalescer(req.K, err);
            if (!co) return false;
            if (pg_coalescer_recall(co, req.Vector.data(), rows.data(), scores.data(), &cnt) != PG_OK) {
                if (err) *err = pg_err("pg_coalescer_recall");
                return false;
            }
        } else if (pg_recall_topk(e->ctx, e->table, req.Vector.data(), 1, req.K, rows.data(), scores.data(), &cnt) != PG_OK) {
            if (err) *err = pg_err("pg_recall_topk");
            return false;
        }
        for (uint32_t i = 0; i < cnt; ++i) {
            out->reply.Retval.push_back(rows[i]);
            out->reply.Scores.push_back(scores[i]);
            out->reply.Labels.push_back(e->IdOfRow(rows[i]));
        }
        return true;
    }
};

// algorithm.IAlgorithm replacing EasModel / TFservingModel for the DNN rank model
struct GpuDnnAlgorithm : algorithm::IAlgorithm {
    Engine* e;
    std::string name;
    std::vector<std::string> outputs;            // empty: one score; else a multi-output model, one DNN3 head model per output
    GpuDnnAlgorithm(Engine* eng, std::string n, std::vector<std::string> outs) : e(eng), name(std::move(n)), outputs(std::move(outs)) {}
    bool Init(const recconf::AlgoConfig&, std::string*) override { return true; }
    bool Score(const pg_model* m, const algorithm::RankRequest& req, const std::vector<uint32_t>& rows, std::vector<float>* scores,
               std::string* err) {
        const uint32_t n = (uint32_t)rows.size();
        const uint32_t off[2] = {0, n};
        scores->resize(n);
        if (!m) { if (err) *err = "dnn: model of " + name + " not loaded"; return false; }
        if (e->coalesce && m == e->model) {
            // one call per 100-item batch and goroutine (rank_service.go:264-289): batched across callers by the library
            pg_coalescer* co = e->SceneCoalescer(0, err);
            if (!co) return false;
            if (pg_coalescer_rank_dnn3(co, req.UserVector.data(), rows.data(), n, scores->data()) != PG_OK) {
                if (err) *err = pg_err("pg_coalescer_rank_dnn3");
                return false;
            }
            return true;
        }
        if (pg_rank_dnn3(e->ctx, m, e->table, req.UserVector.data(), rows.data(), off, 1, scores->data()) != PG_OK) {
            if (err) *err = pg_err("pg_rank_dnn3");
            return false;
        }
        return true;
    }
    bool Run(const algorithm::AlgoData& data, algorithm::AlgoResult* out, std::string* err) override {
        if (data.kind != algorithm::AlgoData::kRank) { if (err) *err = "dnn: invalid request type"; return false; }
        const auto& req = data.rank;
        const uint32_t n = (uint32_t)req.ItemIds.size();
        std::vector<uint32_t> rows(n);
        for (uint32_t i = 0; i < n; ++i)
            if (!e->RowOfId(req.ItemIds[i], &rows[i])) { if (err) *err = "dnn: unknown item id " + req.ItemIds[i]; return false; }
        std::vector<float> scores;
        if (outputs.empty()) {
            if (!Score(e->model, req, rows, &scores, err)) return false;
            out->responses = algorithm::decode::WidenF32(scores.data(), n);             // float32 → float64 widening
            return true;
        }
        // multi-output model (EasyrecResponse.multiValModule, easyrec_response.go:35-70): GetModuleType() = true and a score
        // per output name; RankService writes them as "<algo>_<output>" (rank_service.go:315-319)
        out->responses.assign(n, algorithm::AlgoResponse());
        auto mh = e->named_models.find(name);
        if (mh != e->named_models.end()) {
            // the model as exported: ONE trunk, one head per output (PG_MODEL_DNN3_MULTI) — one gather, one launch
            uint32_t heads = 0;
            pg_model_num_outputs(mh->second, &heads);
            if (heads != outputs.size()) {
                if (err) *err = "dnn: model of " + name + " has " + std::to_string(heads) + " outputs, the algorithm names " +
                                std::to_string(outputs.size());
                return false;
            }
            std::vector<float> planes((size_t)heads * n);
            const uint32_t off[2] = {0, n};
            if (req.UserVector.empty()) { if (err) *err = "dnn: no user vector"; return false; }
            if (pg_rank_dnn3(e->ctx, mh->second, e->table, req.UserVector.data(), rows.data(), off, 1, planes.data()) != PG_OK) {
                if (err) *err = pg_err("pg_rank_dnn3");
                return false;
            }
            for (uint32_t i = 0; i < n; ++i) {
                out->responses[i].multiValModule = true;
                for (uint32_t o = 0; o < heads; ++o) out->responses[i].scoreArr[outputs[o]] = (double)planes[(size_t)o * n + i];
            }
            return true;
        }
        for (const auto& o : outputs) {
            auto it = e->named_models.find(name + "/" + o);
            if (!Score(it == e->named_models.end() ? nullptr : it->second, req, rows, &scores, err)) return false;
            for (uint32_t i = 0; i < n; ++i) {
                out->responses[i].multiValModule = true;
                out->responses[i].scoreArr[o] = (double)scores[i];
            }
        }
        return true;
    }
};

// algorithm.IAlgorithm for the EasyRec flavour (service/rank/algo_data.go:79-86, algorithm/eas/easyrec_request.go:20-73):
// the request names item ids; the per-item context features are device columns (pg_features_*), the user side arrives
// as a dense vector + dictionary-encoded categorical ids.  FM + two-tower predict (pg_rank_fm2t_rows).
struct GpuFm2tAlgorithm : algorithm::IAlgorithm {
    Engine* e;
    std::vector<std::string> columns;
    GpuFm2tAlgorithm(Engine* eng, std::vector<std::string> cols) : e(eng), columns(std::move(cols)) {}
    bool Init(const recconf::AlgoConfig&, std::string*) override { return true; }
    bool Run(const algorithm::AlgoData& data, algorithm::AlgoResult* out, std::string* err) override {
        if (data.kind != algorithm::AlgoData::kRank) { if (err) *err = "fm2t: invalid request type"; return false; }
        if (!e->fm2t || !e->feats) { if (err) *err = "fm2t: model or feature columns not loaded"; return false; }
        const auto& req = data.rank;
        const uint32_t n = (uint32_t)req.ItemIds.size();
        std::vector<uint32_t> rows(n);
        for (uint32_t i = 0; i < n; ++i)
            if (!e->RowOfId(req.ItemIds[i], &rows[i])) { if (err) *err = "fm2t: unknown item id " + req.ItemIds[i]; return false; }
        std::vector<int32_t> cols;
        for (const auto& c : columns) {
            const int idx = pg_features_column_index(e->feats, c.c_str());
            if (idx < 0) { if (err) *err = "fm2t: no feature column " + c; return false; }
            cols.push_back(idx);
        }
        const uint32_t off[2] = {0, n};
        std::vector<float> scores(n);
        if (req.UserVector.size() != e->fm2t_d_user || req.UserFieldIds.size() < e->fm2t_nuf) {
            if (err) *err = "fm2t: user vector / field ids do not match the model";
            return false;
        }
        if (e->coalesce && columns == e->fm2t_columns) {
            // 50 calls of 100 items per request, from as many goroutines (rank_service.go:264-289): one launch for all of them
            pg_coalescer* co = e->SceneCoalescer(0, err);
            if (!co) return false;
            if (pg_coalescer_rank_fm2t(co, req.UserVector.data(), req.UserFieldIds.data(), rows.data(), n, scores.data()) != PG_OK) {
                if (err) *err = pg_err("pg_coalescer_rank_fm2t");
                return false;
            }
        } else if (pg_rank_fm2t_rows(e->ctx, e->fm2t, e->feats, cols.data(), req.UserVector.data(), req.UserFieldIds.data(), rows.data(), off, 1,
                                     scores.data()) != PG_OK) {
            if (err) *err = pg_err("pg_rank_fm2t_rows");
            return false;
        }
        out->responses = algorithm::decode::WidenF32(scores.data(), n);
        return true;
    }
};

// algorithm.IAlgorithm of a vector model serving an OnlineVectorRecall: user features → user embedding → its
// FaissNeighNum nearest items (torchrecEmbeddingItemsResponseFunc, easyrec_response.go:700-734)
struct GpuOnlineVectorAlgorithm : algorithm::IAlgorithm {
    Engine* e;
    explicit GpuOnlineVectorAlgorithm(Engine* eng) : e(eng) {}
    bool Init(const recconf::AlgoConfig&, std::string*) override { return true; }
    bool Run(const algorithm::AlgoData& data, algorithm::AlgoResult* out, std::string* err) override {
        if (data.kind != algorithm::AlgoData::kEmbedding) { if (err) *err = "online vector: invalid request type"; return false; }
        if (!e->fm2t || !e->item_emb) { if (err) *err = "online vector: vector model or item-embedding table not loaded"; return false; }
        const uint32_t k = (uint32_t)std::max(data.emb.FaissNeighNum, 0);
        if (k == 0) return true;
        std::vector<uint64_t> rows(k);
        std::vector<float> scores(k);
        uint32_t cnt = 0;
        if (data.emb.UserVector.size() != e->fm2t_d_user) { if (err) *err = "online vector: user vector width does not match the model"; return false; }
        if (e->coalesce) {
            pg_coalescer* co = e->OnlineCoalescer(k, err);
            if (!co) return false;
            if (pg_coalescer_online_recall(co, data.emb.UserVector.data(), rows.data(), scores.data(), &cnt) != PG_OK) {
                if (err) *err = pg_err("pg_coalescer_online_recall");
                return false;
            }
        } else if (pg_online_vector_recall(e->ctx, e->fm2t, e->item_emb, data.emb.UserVector.data(), 1, k, rows.data(), scores.data(), &cnt) != PG_OK) {
            if (err) *err = pg_err("pg_online_vector_recall");
            return false;
        }
        for (uint32_t i = 0; i < cnt; ++i) out->embeddingItems.push_back({e->IdOfRow(rows[i]), (double)scores[i]});
        return true;
    }
};

// recall.Recall with the body of VectorRecall.GetCandidateItems (vector_recall.go:32-123, cache omitted)
struct GpuVectorRecall : recall::Recall, recall::ICloneRecall {
    Engine* e;
    recconf::RecallConfig conf;
    std::shared_ptr<cache::Cache> cache_;                      // BaseRecall.cache (recall.go:123-137)
    int cacheTime = 1800;
    std::mutex cloneMu;
    std::map<std::string, std::shared_ptr<recall::Recall>> cloneInstances;
    GpuVectorRecall(Engine* eng, recconf::RecallConfig c) : e(eng), conf(std::move(c)) {
        if (!conf.CacheAdapter.empty()) {
            std::string cerr;
            cache_ = cache::NewCache(conf.CacheAdapter, conf.CacheConfig, &cerr);   // error → logged, no cache
            if (conf.CacheTime > 0) cacheTime = conf.CacheTime;
        }
    }
    std::string GetRecallName() const override { return conf.Name; }
    // cloneWithBuilder (recall.go:167-214): instance-local cache keyed by the params' canonical JSON (Go
    // marshals maps with sorted keys and takes an md5 of that; the text itself is the key here), params
    // unmarshalled over a zero RecallConfig, Name forced to the original
    std::shared_ptr<recall::Recall> CloneWithConfig(const json::Value& params) override {
        if (params.type != json::Value::Object) return nullptr;
        std::string key;
        for (const auto& kv : params.obj) {                    // std::map iterates in key order
            key += kv.first + "=";
            key += kv.second.type == json::Value::String ? kv.second.str : json::NumToString(kv.second.num);
            key += ";";
        }
        std::lock_guard<std::mutex> g(cloneMu);
        auto it = cloneInstances.find(key);
        if (it != cloneInstances.end()) return it->second;
        recconf::RecallConfig c;
        c.Name = conf.Name;
        c.RecallType = params.s("RecallType"); c.RecallAlgo = params.s("RecallAlgo");
        c.ItemType = params.s("ItemType"); c.CacheAdapter = params.s("CacheAdapter");
        c.CacheConfig = params.s("CacheConfig"); c.CachePrefix = params.s("CachePrefix");
        c.RecallCount = (int)params.n("RecallCount"); c.CacheTime = (int)params.n("CacheTime");
        auto inst = std::make_shared<GpuVectorRecall>(e, c);
        cloneInstances[key] = inst;
        return inst;
    }
    std::vector<module::ItemPtr> GetCandidateItems(module::User* user, context::RecommendContext*) override {
        std::vector<module::ItemPtr> ret;
        std::string value, err;
        if (cache_) {                                                                   // vector_recall.go:35-58
            const cache::Value v = cache_->Get(conf.CachePrefix + user->Id);
            if (v.kind == cache::Value::kBytes) {                                       // `cacheRet.([]uint8)`
                if (recall::ParseCacheString(v.data, conf.Name, conf.ItemType, &ret, &err)) return ret;
                ret.clear();
            }
        }
        if (!e->user_vectors.VectorString(user->Id, &value, &err)) return ret;        // logged, empty result
        algorithm::AlgoData data;
        data.kind = algorithm::AlgoData::kVector;
        data.vec.K = (uint32_t)conf.RecallCount;
        data.vec.Vector = recall::ParseVectorString(value);
        if (data.vec.Vector.empty()) return ret;                                       // "user Vector empty"
        algorithm::AlgoResult result;
        if (!e->algorithms.Run(conf.RecallAlgo, data, &result, &err)) return ret;      // logged, empty result
        for (size_t i = 0; i < result.reply.Labels.size(); ++i) {
            auto item = std::make_shared<module::Item>(result.reply.Labels[i]);
            item->RetrieveId = conf.Name;
            item->ItemType = conf.ItemType;
            item->Score = (double)result.reply.Scores[i];
            ret.push_back(item);
        }
        if (cache_ && !ret.empty())                                                     // :103-120 (async in Go)
            cache_->Put(conf.CachePrefix + user->Id, recall::FormatCacheString(ret, conf.Name), cacheTime);
        return ret;
    }
};

// recall.Recall with the body of I2IVectorRecall.GetCandidateItems (item_2_item_vector_racall.go:51-152): the trigger is
// the request's "item_id" parameter, its embedding (dao.VectorString) the query
struct GpuI2IVectorRecall : recall::Recall {
    Engine* e;
    recconf::RecallConfig conf;
    GpuI2IVectorRecall(Engine* eng, recconf::RecallConfig c) : e(eng), conf(std::move(c)) {}
    std::vector<module::ItemPtr> GetCandidateItems(module::User*, context::RecommendContext* ctx) override {
        std::vector<module::ItemPtr> ret;
        uint32_t row = 0;
        if (!ctx || !e->RowOfId(ctx->GetParameter("item_id"), &row)) return ret;       // VectoryEmptyError: logged, empty
        const uint32_t k = (uint32_t)std::max(conf.RecallCount, 0);
        if (k == 0) return ret;
        std::vector<uint64_t> rows(k);
        std::vector<float> scores(k);
        uint32_t cnt = 0;
        if (e->coalesce) {
            std::string cerr;
            pg_coalescer* co = e->SceneCoalescer(k, &cerr);
            if (!co || pg_coalescer_i2i_recall(co, row, rows.data(), scores.data(), &cnt) != PG_OK) return ret;
        } else if (pg_i2i_recall(e->ctx, e->table, &row, 1, e->table, k, rows.data(), scores.data(), &cnt) != PG_OK) {
            return ret;
        }
        for (uint32_t i = 0; i < cnt; ++i) {
            auto item = std::make_shared<module::Item>(e->IdOfRow(rows[i]));
            item->RetrieveId = conf.Name;
            item->ItemType = conf.ItemType;
            item->Score = (double)scores[i];                                            // `distance float64` (:130-139)
            ret.push_back(item);
        }
        return ret;
    }
};

// recall.Recall with the bodies of HologresVectorRecall.GetCandidateItems (service/recall/hologres_vector_recall.go:94-206: "ORDER
// BY pm_approx_inner_product_distance(emb, $1) desc LIMIT RecallCount", :23) and HologresVectorRecallV2.GetCandidateItems
// (hologres_vector_recall_v2.go:96-206: "ORDER BY pm_approx_squared_euclidean_distance(emb, $1) LIMIT RecallCount", :23): the user's
// embedding from the VectorDao (handed to the SQL as it is — "v1,v2,…" or "{v1,v2,…}", module/vector_hologres_dao.go:67-114; the
// libsvm form of the other DAOs is accepted too), the RecallCount items of largest inner product (descending) / smallest squared
// Euclidean distance (ascending), Score = distance (v1 :175-183, v2 :181-189), among the rows the WhereClause admits (:56-61).
// (The user-vector cache of :100-115 is the VectorDao's business here; the result cache of :118-143 / :191-204 is GpuVectorRecall's.)
struct GpuHologresVectorRecall : recall::Recall {
    Engine* e;
    recconf::RecallConfig conf;
    bool l2;
    GpuHologresVectorRecall(Engine* eng, recconf::RecallConfig c, bool squared_euclidean = true) : e(eng), conf(std::move(c)), l2(squared_euclidean) {}
    static std::vector<float> ParseEmbedding(const std::string& s) {
        if (s.find(':') != std::string::npos) return recall::ParseVectorString(s);
        std::vector<float> out;
        size_t p = 0;
        while (p < s.size()) {
            while (p < s.size() && (s[p] == '{' || s[p] == '}' || s[p] == ',' || s[p] == ' ')) ++p;
            if (p >= s.size()) break;
            char* end = nullptr;
            const double d = strtod(s.c_str() + p, &end);
            if (end == s.c_str() + p) break;
            out.push_back((float)d);
            p = (size_t)(end - s.c_str());
        }
        return out;
    }
    std::vector<module::ItemPtr> GetCandidateItems(module::User* user, context::RecommendContext*) override {
        std::vector<module::ItemPtr> ret;
        std::string value, err;
        if (!e->user_vectors.VectorString(user->Id, &value, &err) || value.empty()) return ret;      // (:105-110) logged unless VectoryEmptyError
        user->Properties[conf.Name + "_embedding"] = json::Value::Str(value);                        // user.AddProperty (:111)
        const std::vector<float> vec = ParseEmbedding(value);
        const uint32_t k = (uint32_t)std::max(conf.RecallCount, 0);
        if (k == 0 || vec.size() != e->dim) return ret;                                               // the SQL would fail: logged, empty (:170-176)
        std::vector<uint64_t> rows(k);
        std::vector<float> dist(k);
        uint32_t cnt = 0;
        if (conf.WhereOp >= 0) {
            // the admitted rows as a view of the table, built by the first request of a table generation; concurrent requests
            // share its passes through the view's own coalescer
            const Engine::FilterView* fv = e->ViewFor(conf, k);
            if (!fv || fv->empty) return ret;               // unknown column: the SQL would fail (logged, empty :170-176); nothing admitted
            int rc;
            if (fv->co)
                rc = l2 ? pg_coalescer_recall_l2(fv->co, vec.data(), rows.data(), dist.data(), &cnt)
                        : pg_coalescer_recall(fv->co, vec.data(), rows.data(), dist.data(), &cnt);
            else if (fv->view)
                rc = l2 ? pg_recall_topk_l2(e->ctx, fv->view, vec.data(), 1, k, rows.data(), dist.data(), &cnt)
                        : pg_recall_topk(e->ctx, fv->view, vec.data(), 1, k, rows.data(), dist.data(), &cnt);
            else                                            // the view could not be built (memory): filter per call
                rc = pg_recall_topk_where(e->ctx, e->table, e->feats, pg_features_column_index(e->feats, conf.WhereColumn.c_str()), conf.WhereOp,
                                          conf.WhereValue, l2 ? 1 : 0, vec.data(), 1, k, rows.data(), dist.data(), &cnt);
            if (rc != PG_OK) return ret;
        } else if (e->coalesce) {                           // concurrent requests share the exact pass
            std::string cerr;
            pg_coalescer* co = e->SceneCoalescer(k, &cerr);
            if (!co) return ret;
            const int rc = l2 ? pg_coalescer_recall_l2(co, vec.data(), rows.data(), dist.data(), &cnt)
                              : pg_coalescer_recall(co, vec.data(), rows.data(), dist.data(), &cnt);
            if (rc != PG_OK) return ret;
        } else if ((l2 ? pg_recall_topk_l2(e->ctx, e->table, vec.data(), 1, k, rows.data(), dist.data(), &cnt)
                       : pg_recall_topk(e->ctx, e->table, vec.data(), 1, k, rows.data(), dist.data(), &cnt)) != PG_OK) {
            return ret;
        }
        for (uint32_t i = 0; i < cnt; ++i) {
            auto item = std::make_shared<module::Item>(e->IdOfRow(rows[i]));
            item->RetrieveId = conf.Name;
            item->ItemType = conf.ItemType;
            item->Score = (double)dist[i];                                                            // `item.Score = distance` (:189)
            ret.push_back(item);
        }
        return ret;
    }
};

// recall.Recall that returns the FINISHED page: recall → DNN rank → RankScore → ItemRankScore sort all happen where the
// rows live, one single-request call into the library's coalescer (pg_coalescer_recommend), and only ctx.Size items are
// materialised on the host.  A scene served this way names it as its only recall and leaves RankConf / SortNames empty
// (the default ItemRankScore sort keeps the order: Score is the fused score).  The reference has no such plug-in — it
// is what its Recall interface permits (recall.go:18-20) once the stages behind it are local.
struct GpuPageRecall : recall::Recall {
    Engine* e;
    recconf::RecallConfig conf;
    GpuPageRecall(Engine* eng, recconf::RecallConfig c) : e(eng), conf(std::move(c)) {}
    std::vector<module::ItemPtr> GetCandidateItems(module::User* user, context::RecommendContext* ctx) override {
        std::vector<module::ItemPtr> ret;
        std::string value, err;
        if (!e->user_vectors.VectorString(user->Id, &value, &err)) return ret;          // logged, empty result
        const std::vector<float> vec = recall::ParseVectorString(value);
        if (vec.size() != e->dim) return ret;                                           // (logged) the library copies exactly dim floats
        pg_coalescer* co = e->PageCoalescer(conf, &err);
        if (!co) return ret;                                                            // logged, empty result
        const uint32_t size = (uint32_t)std::max(1, std::min(ctx ? ctx->Size : 10, std::min(conf.RecallCount, 1000)));
        std::vector<uint64_t> rows(size);
        std::vector<float> rec(size), rnk(size);
        std::vector<double> fused(size);
        uint32_t cnt = 0;
        if (pg_coalescer_recommend(co, vec.data(), size, rows.data(), rec.data(), rnk.data(), fused.data(), &cnt) != PG_OK) return ret;
        for (uint32_t i = 0; i < cnt; ++i) {
            auto item = std::make_shared<module::Item>(e->IdOfRow(rows[i]));
            item->RetrieveId = conf.Name;
            item->ItemType = conf.ItemType;
            item->Score = fused[i];
            item->AddProperty("recall_score", json::Value::Num((double)rec[i]));                        // what ${current_score} leaves behind
            item->AddAlgoScore(conf.RankVar, (double)rnk[i]);
            ret.push_back(item);
        }
        return ret;
    }
};

// recall.Recall with the body of OnlineVectorRecall.GetCandidateItems (online_vector_recall.go:73-155, cache omitted):
// user features → PBRequest{FaissNeighNum = recallCount} → algorithm.Run(recallAlgo) → embedding items → Items
struct GpuOnlineVectorRecall : recall::Recall {
    Engine* e;
    recconf::RecallConfig conf;
    GpuOnlineVectorRecall(Engine* eng, recconf::RecallConfig c) : e(eng), conf(std::move(c)) {}
    std::vector<module::ItemPtr> GetCandidateItems(module::User* user, context::RecommendContext*) override {
        std::vector<module::ItemPtr> ret;
        std::string value, err;
        if (!e->user_vectors.VectorString(user->Id, &value, &err)) return ret;
        algorithm::AlgoData data;
        data.kind = algorithm::AlgoData::kEmbedding;
        data.emb.UserVector = recall::ParseVectorString(value);
        data.emb.FaissNeighNum = conf.RecallCount;
        algorithm::AlgoResult result;
        if (!e->algorithms.Run(conf.RecallAlgo, data, &result, &err)) return ret;       // logged, empty result
        // only the TorchRec vector flavours carry embedding items (:118-133)
        if (conf.VectorAlgoType != "torchrec_tdm" && conf.VectorAlgoType != "torchrec_vector") return ret;
        for (const auto& info : result.embeddingItems) {
            auto item = std::make_shared<module::Item>(info.ItemId);
            item->Score = info.Score;
            item->RetrieveId = conf.Name;
            ret.push_back(item);
        }
        if (conf.RecallCount > 0 && (int)ret.size() > conf.RecallCount) ret.resize((size_t)conf.RecallCount);
        return ret;
    }
};

// recall.Recall with the body of OnlineHologresVectorRecall.GetCandidateItems (online_hologres_vector_recall.go:105-237, result
// cache omitted): the user's features → the vector model's USER TOWER on the device (the reference asks a PAI-EAS model for
// EasyrecUserEmbResponse.GetUserEmb, :120-131, and keeps it in userVectorCache, :109-112,133) → the RecallCount rows of the
// Hologres vector table with the largest inner product (`pm_approx_inner_product_distance(...) ORDER BY distance desc`, :27),
// restricted by HologresVectorConf.WhereClause (:59-63) — Item.Score = distance, RetrieveId = the recall's name (:208-210).
// The vector table is the engine's item-embedding table (UserDefineConfs.pairec_gpu.OnlineVector), the clause's column an
// int32 feature column keyed by its rows.  One embedding per user (the reference splits GetUserEmb on "|" for multi-interest
// models and gives every part RecallCount / parts rows, :188-196).
struct GpuOnlineHologresVectorRecall : recall::Recall {
    Engine* e;
    recconf::RecallConfig conf;
    struct CachedEmb { std::vector<float> v; std::chrono::steady_clock::time_point used; };
    std::mutex mu;
    std::map<std::string, CachedEmb> userVectorCache;              // cache.WithMaximumSize(10000), expire-after-access cacheTime + 10 s (:74-77)
    GpuOnlineHologresVectorRecall(Engine* eng, recconf::RecallConfig c) : e(eng), conf(std::move(c)) {}
    bool UserEmbedding(const std::string& uid, const std::vector<float>& uv, std::vector<float>* emb) {
        const auto now = std::chrono::steady_clock::now();
        const auto ttl = std::chrono::seconds(std::max(conf.CacheTime, 0) + 10);
        {
            std::lock_guard<std::mutex> g(mu);
            auto it = userVectorCache.find(conf.CachePrefix + uid);
            if (it != userVectorCache.end() && now - it->second.used <= ttl) {
                it->second.used = now;
                *emb = it->second.v;
                return true;
            }
        }
        emb->assign(e->dim_item_emb(), 0.0f);
        if (pg_fm2t_user_embedding(e->ctx, e->fm2t, uv.data(), 1, emb->data()) != PG_OK) return false;     // logged, empty (:123-125)
        std::lock_guard<std::mutex> g(mu);
        if (userVectorCache.size() >= 10000) userVectorCache.erase(userVectorCache.begin());
        userVectorCache[conf.CachePrefix + uid] = CachedEmb{*emb, now};
        return true;
    }
    std::vector<module::ItemPtr> GetCandidateItems(module::User* user, context::RecommendContext*) override {
        std::vector<module::ItemPtr> ret;
        std::string value, err;
        if (!e->fm2t || !e->item_emb) return ret;                                                  // no model / vector table: logged, empty
        if (!e->user_vectors.VectorString(user->Id, &value, &err)) return ret;
        const std::vector<float> uv = recall::ParseVectorString(value);
        const uint32_t k = (uint32_t)std::max(conf.RecallCount, 0);
        if (k == 0 || uv.size() != e->fm2t_d_user) return ret;
        std::vector<uint64_t> rows(k);
        std::vector<float> dist(k);
        uint32_t cnt = 0;
        int rc;
        if (conf.WhereOp >= 0) {
            std::vector<float> emb;
            if (!UserEmbedding(user->Id, uv, &emb)) return ret;
            const Engine::FilterView* fv = e->ViewFor(conf, k, e->item_emb);
            if (!fv || fv->empty) return ret;                   // unknown column: the SQL would fail (logged, empty); nothing admitted
            if (fv->co) rc = pg_coalescer_recall(fv->co, emb.data(), rows.data(), dist.data(), &cnt);
            else if (fv->view) rc = pg_recall_topk(e->ctx, fv->view, emb.data(), 1, k, rows.data(), dist.data(), &cnt);
            else rc = pg_recall_topk_where(e->ctx, e->item_emb, e->feats, pg_features_column_index(e->feats, conf.WhereColumn.c_str()), conf.WhereOp,
                                           conf.WhereValue, 0, emb.data(), 1, k, rows.data(), dist.data(), &cnt);
        } else if (e->coalesce) {                               // tower + search of concurrent requests in one pass
            pg_coalescer* co = e->OnlineCoalescer(k, &err);
            if (!co) return ret;
            rc = pg_coalescer_online_recall(co, uv.data(), rows.data(), dist.data(), &cnt);
        } else {
            rc = pg_online_vector_recall(e->ctx, e->fm2t, e->item_emb, uv.data(), 1, k, rows.data(), dist.data(), &cnt);
        }
        if (rc != PG_OK) return ret;
        for (uint32_t i = 0; i < cnt; ++i) {
            auto item = std::make_shared<module::Item>(e->IdOfRow(rows[i]));
            item->RetrieveId = conf.Name;
            item->Score = (double)dist[i];
            ret.push_back(item);
        }
        return ret;
    }
};

// MockRecall (service/recall/mock_recall.go:27-41): recallCount random ids with random scores
struct MockRecall : recall::Recall {
    recconf::RecallConfig conf;
    explicit MockRecall(recconf::RecallConfig c) : conf(std::move(c)) {}
    std::vector<module::ItemPtr> GetCandidateItems(module::User*, context::RecommendContext*) override {
        std::vector<module::ItemPtr> ret;
        uint64_t s = 0x9E3779B97F4A7C15ull;
        while ((int)ret.size() < conf.RecallCount) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            auto item = std::make_shared<module::Item>(std::to_string((uint32_t)(s >> 32)));
            item->RetrieveId = conf.Name;
            item->Score = (double)(s >> 11) * (1.0 / 9007199254740992.0);
            ret.push_back(item);
        }
        return ret;
    }
};

bool sort_items(Engine* e, sort::SortData* d, bool desc, std::string* err) {
    const uint32_t n = (uint32_t)d->Data.size();
    if (n == 0) return true;
    std::vector<double> s(n);
    for (uint32_t i = 0; i < n; ++i) s[i] = d->Data[i]->Score;
    const uint32_t seg[2] = {0, n};
    std::vector<uint32_t> order(n);
    if (pg_sort_scores(e->ctx, s.data(), seg, 1, desc ? 1 : 0, order.data()) != PG_OK) {
        if (err) *err = pg_err("pg_sort_scores");
        return false;                                   // caller ignores the error; Data left untouched
    }
    std::vector<module::ItemPtr> out(n);
    for (uint32_t i = 0; i < n; ++i) out[i] = d->Data[order[i]];
    d->Data.swap(out);
    return true;
}
struct GpuItemRankScoreSort : sort::ISort {              // sort/item_rank_score.go:26-32 (descending)
    Engine* e;
    explicit GpuItemRankScoreSort(Engine* eng) : e(eng) {}
    bool Sort(sort::SortData* d, std::string* err) override { return sort_items(e, d, true, err); }
};
struct GpuItemScoreSort : sort::ISort {                  // sort/item_score.go:36-41 (ascending)
    Engine* e;
    explicit GpuItemScoreSort(Engine* eng) : e(eng) {}
    bool Sort(sort::SortData* d, std::string* err) override { return sort_items(e, d, false, err); }
};
struct GpuDPPSort : sort::ISort {                        // sort/dpp_sort.go:108-351 (embedding table = item table)
    Engine* e;
    recconf::DPPSortConfig conf;
    GpuDPPSort(Engine* eng, recconf::DPPSortConfig c) : e(eng), conf(std::move(c)) {}
    // doSort (:271-351): experiment parameters override the config (:275-278,374,382)
    bool DoSort(std::vector<module::ItemPtr>* itemsp, sort::SortData* d, std::string* err) {
        auto& items = *itemsp;
        if (items.empty()) return true;
        const int size = d->Context ? d->Context->Size : 10;
        const context::RecommendContext none;
        const context::RecommendContext& cx = d->Context ? *d->Context : none;
        int window = (int)cx.GetInt("dpp_window_size", conf.WindowSize > 0 ? conf.WindowSize : 10);
        const int candidateCnt = (int)cx.GetInt("dpp_candidate_count", conf.CandidateCount);
        const double minScorePercent = cx.GetFloat("dpp_min_score_percent", conf.MinScorePercent);
        const double alpha = cx.GetFloat("dpp_alpha", conf.Alpha);
        const int doNorm = (int)cx.GetInt("dpp_norm_relevance_score", 0);
        if ((candidateCnt > 0 || minScorePercent > 0) && (int)items.size() > size) {   // :280-300
            sort::SortData tmp = *d;
            tmp.Data = items;
            if (!sort_items(e, &tmp, true, err)) return false;
            items.swap(tmp.Data);
            if (candidateCnt > 0) {
                const size_t cnt = (size_t)std::max(size, candidateCnt);
                if (cnt < items.size()) items.resize(cnt);
            }
            if (minScorePercent > 0 && (int)items.size() > size) {
                size_t idx = (size_t)size;
                const double mx = items[0]->Score;
                for (; idx < items.size(); ++idx)
                    if (items[idx]->Score / mx < minScorePercent) break;
                items.resize(idx);
            }
        }
        const uint32_t n = (uint32_t)items.size();
        std::vector<uint32_t> rows(n);
        std::vector<double> rel(n), used(n);
        for (uint32_t i = 0; i < n; ++i) {
            if (!e->RowOfId(items[i]->Id, &rows[i])) { if (err) *err = "dpp: unknown item id"; return false; }
            rel[i] = items[i]->Score;
        }
        pg_dpp_options o;
        memset(&o, 0, sizeof o);
        o.alpha = alpha;
        o.topn = (uint32_t)std::max(size, 0);
        o.window = (uint32_t)std::max(window, 0);
        o.normalize_emb = conf.NormalizeEmb ? 1 : 0;
        o.ensure_pos_similarity = conf.EnsurePositiveSim ? 1 : 0;
        o.norm_relevance_score = (doNorm == 1 || doNorm == 2) ? doNorm : 0;
        o.has_table = 1;
        std::vector<uint32_t> idx((size_t)std::max(size, 1));
        uint32_t cnt = 0;
        int rc;
        std::string cerr;                               // (a failed coalescer falls back to the direct call: not this call's error)
        pg_coalescer* co = e->coalesce ? e->SceneCoalescer(0, &cerr) : nullptr;
        // SortService.Sort runs once per request, requests overlap (sort/sort.go:65-125): equal-shaped DPP calls share a launch
        if (co && n <= 1024) rc = pg_coalescer_dpp(co, rows.data(), rel.data(), n, &o, nullptr, idx.data(), &cnt, used.data());
        else rc = pg_dpp_ex(e->ctx, e->table, rows.data(), rel.data(), n, &o, nullptr, idx.data(), &cnt, used.data());
        if (rc == PG_ERR_ARITH) return true;            // "all item score is zero": KernelMatrix errs, items stay as they are (:314-317)
        if (rc != PG_OK) {
            if (err) *err = pg_err("pg_dpp");
            return false;
        }
        for (uint32_t i = 0; i < n; ++i) items[i]->AddAlgoScore("dpp_relevance_score", used[i]);     // :410
        std::vector<module::ItemPtr> out;
        for (uint32_t i = 0; i < cnt; ++i) out.push_back(items[idx[i]]);
        items.swap(out);
        return true;
    }
    bool Sort(sort::SortData* d, std::string* err) override {
        auto& items = d->Data;
        if (items.empty()) return true;
        if (conf.AbortRunCount > 0 && (int)items.size() <= conf.AbortRunCount)      // :127-132
            return sort_items(e, d, true, err);
        if (!conf.FilterRetrieveIds.empty()) {                                       // :134-160
            std::vector<module::ItemPtr> backup, selected;
            for (auto& it : items) {
                const bool f = std::find(conf.FilterRetrieveIds.begin(), conf.FilterRetrieveIds.end(), it->RetrieveId) !=
                               conf.FilterRetrieveIds.end();
                (f ? backup : selected).push_back(it);
            }
            if (!DoSort(&selected, d, err)) return false;
            selected.insert(selected.end(), backup.begin(), backup.end());
            items.swap(selected);
            return true;
        }
        return DoSort(&items, d, err);
    }
};

// AlgoScoreSort (sort/algo_score_sort.go:38-66): descending by SortByField unless the best Item.Score exceeds
// SwitchThreshold, then by current_score.  (The reference's comparator assigns a failed RIGHT lookup's fallback to the
// left score — `iScore = items[j].Score` — which only matters when the field is missing; here a missing field falls
// back to the item's own score on both sides.)
struct GpuAlgoScoreSort : sort::ISort {
    Engine* e;
    std::string sortByField;
    double switchThreshold;
    GpuAlgoScoreSort(Engine* eng, const recconf::SortConfig& c)
        : e(eng), sortByField(c.SortByField.empty() ? "current_score" : c.SortByField), switchThreshold(c.SwitchThreshold) {}
    bool Sort(sort::SortData* d, std::string* err) override {
        const uint32_t n = (uint32_t)d->Data.size();
        if (n == 0) return true;
        double maxScore = -1e300;
        for (const auto& it : d->Data) maxScore = std::max(maxScore, it->Score);
        const std::string field = maxScore > switchThreshold ? "current_score" : sortByField;
        std::vector<double> key(n);
        for (uint32_t i = 0; i < n; ++i)
            if (!d->Data[i]->FloatExprData(field, &key[i])) key[i] = d->Data[i]->Score;
        const uint32_t seg[2] = {0, n};
        std::vector<uint32_t> order(n);
        if (pg_sort_scores(e->ctx, key.data(), seg, 1, 1, order.data()) != PG_OK) {
            if (err) *err = pg_err("pg_sort_scores");
            return false;
        }
        std::vector<module::ItemPtr> out(n);
        for (uint32_t i = 0; i < n; ++i) out[i] = d->Data[order[i]];
        d->Data.swap(out);
        return true;
    }
};

// CustomFieldSort (sort/custom_field_sort.go:21-73): by Item.FloatExprData(SortByField) — an algo score, else a property, "current_score"
// = Item.Score — falling back to the item's own Score where the field is missing; "asc" or (default, and for anything else) "desc".
// The order is the device sort's (ties by position, where Go's sort.Slice leaves them unspecified).
struct GpuCustomFieldSort : sort::ISort {
    Engine* e;
    std::string sortByField, sortOrder;
    GpuCustomFieldSort(Engine* eng, const recconf::SortConfig& c) : e(eng), sortByField(c.SortByField.empty() ? "current_score" : c.SortByField) {
        sortOrder = c.SortOrder;
        std::transform(sortOrder.begin(), sortOrder.end(), sortOrder.begin(), ::tolower);
        if (sortOrder != "asc" && sortOrder != "desc") sortOrder = "desc";
    }
    bool Sort(sort::SortData* d, std::string* err) override {
        const uint32_t n = (uint32_t)d->Data.size();
        if (n == 0) return true;
        std::vector<double> key(n);
        for (uint32_t i = 0; i < n; ++i)
            if (!d->Data[i]->FloatExprData(sortByField, &key[i])) key[i] = d->Data[i]->Score;
        const uint32_t seg[2] = {0, n};
        std::vector<uint32_t> order(n);
        if (pg_sort_scores(e->ctx, key.data(), seg, 1, sortOrder == "asc" ? 0 : 1, order.data()) != PG_OK) {
            if (err) *err = pg_err("pg_sort_scores");
            return false;
        }
        std::vector<module::ItemPtr> out(n);
        for (uint32_t i = 0; i < n; ++i) out[i] = d->Data[order[i]];
        d->Data.swap(out);
        return true;
    }
};

struct GpuSSDSort : sort::ISort {                        // sort/ssd_sort.go:110-343 (embedding table = item table)
    Engine* e;
    recconf::SSDSortConfig conf;
    GpuSSDSort(Engine* eng, recconf::SSDSortConfig c) : e(eng), conf(std::move(c)) {}
    // doSort (:296-343): score-descending order, candidate trimming, then SSDWithSlidingWindow
    bool DoSort(std::vector<module::ItemPtr>* items, sort::SortData* d, std::string* err) {
        if (items->empty()) return true;
        const int size = d->Context ? d->Context->Size : 10;
        sort::SortData tmp = *d;
        tmp.Data = *items;
        if (!sort_items(e, &tmp, true, err)) return false;
        items->swap(tmp.Data);
        // experiment parameters override the config (ssd_sort.go:301-309,354-356,364)
        const context::RecommendContext none;
        const context::RecommendContext& cx = d->Context ? *d->Context : none;
        const double gamma = cx.GetFloat("ssd_gamma", conf.Gamma);
        if (gamma == 0) return true;                                                // :302-305
        const int candidateCnt = (int)cx.GetInt("ssd_candidate_count", conf.CandidateCount);
        const double minScorePercent = cx.GetFloat("ssd_min_score_percent", conf.MinScorePercent);
        const int windowSize = (int)cx.GetInt("ssd_window_size", conf.WindowSize);
        const int doNorm = (int)cx.GetInt("ssd_norm_quality_score", 0);
        if ((candidateCnt > 0 || minScorePercent > 0) && (int)items->size() > size) {
            if (candidateCnt > 0) {
                const size_t cnt = (size_t)std::max(size, candidateCnt);
                if (cnt < items->size()) items->resize(cnt);
            }
            if (minScorePercent > 0 && (int)items->size() > size) {
                size_t idx = (size_t)size;
                const double mx = (*items)[0]->Score;
                for (; idx < items->size(); ++idx)
                    if ((*items)[idx]->Score / mx < minScorePercent) break;
                items->resize(idx);
            }
        }
        const uint32_t n = (uint32_t)items->size();
        std::vector<uint32_t> rows(n);
        std::vector<double> rel(n);
        for (uint32_t i = 0; i < n; ++i) {
            if (!e->RowOfId((*items)[i]->Id, &rows[i])) { if (err) *err = "ssd: unknown item id"; return false; }
            rel[i] = (*items)[i]->Score;
        }
        std::vector<uint32_t> idx(n);
        std::vector<double> quality(n);
        uint32_t cnt = 0;
        // SortService.Sort runs once per request, requests overlap (sort/sort.go:65-125): equal-shaped SSD calls of concurrent
        // requests share one launch (every request its own workgroups and barrier); other shapes take the direct call
        int src = PG_ERR_UNSUPPORTED;
        std::string cerr;                               // (a failed coalescer falls back to the direct call: not this call's error)
        if (e->coalesce && n <= 1024)
            if (pg_coalescer* co = e->SceneCoalescer(0, &cerr))
                src = pg_coalescer_ssd(co, rows.data(), rel.data(), n, gamma, (uint32_t)std::max(size, 0), (uint32_t)std::max(windowSize, 0),
                                       conf.NormalizeEmb ? 1 : 0, conf.EnsurePositiveSim ? 1 : 0, (doNorm == 1 || doNorm == 2) ? doNorm : 0,
                                       conf.UseSSDStar ? 1 : 0, idx.data(), &cnt, quality.data());
        if (src == PG_ERR_UNSUPPORTED)
            src = pg_ssd(e->ctx, e->table, rows.data(), rel.data(), n, gamma, (uint32_t)std::max(size, 0),
                         (uint32_t)std::max(windowSize, 0), conf.NormalizeEmb ? 1 : 0, conf.EnsurePositiveSim ? 1 : 0,
                         (doNorm == 1 || doNorm == 2) ? doNorm : 0, conf.UseSSDStar ? 1 : 0, idx.data(), &cnt, quality.data());
        if (src != PG_OK) {
            if (err) *err = pg_err("pg_ssd");
            return false;
        }
        if ((doNorm == 1 || doNorm == 2) && cnt != n)        // :372,385 (not on the "all zeros" bail-out)
            for (uint32_t i = 0; i < n; ++i) (*items)[i]->AddAlgoScore("ssd_quality_score", quality[i]);
        std::vector<module::ItemPtr> out;
        for (uint32_t i = 0; i < cnt; ++i) out.push_back((*items)[idx[i]]);
        items->swap(out);
        return true;
    }
    bool Sort(sort::SortData* d, std::string* err) override {
        auto& items = d->Data;
        if (items.empty()) return true;
        if (conf.AbortRunCount > 0 && (int)items.size() <= conf.AbortRunCount)      // :129-134
            return sort_items(e, d, true, err);
        std::vector<std::string> filterIds;                                          // :136-152
        if (d->Context && d->Context->HasExperiment())
            for (const auto& v : d->Context->ExperimentParamsJson.at("ssd_filter_retrieve_ids").arr)
                if (v.type == json::Value::String) filterIds.push_back(v.str);
        if (filterIds.empty()) filterIds = conf.FilterRetrieveIds;
        if (!filterIds.empty()) {                                                    // :155-171
            std::vector<module::ItemPtr> backup, selected;
            for (auto& it : items) {
                const bool f = std::find(filterIds.begin(), filterIds.end(), it->RetrieveId) != filterIds.end();
                (f ? backup : selected).push_back(it);
            }
            if (!DoSort(&selected, d, err)) return false;
            selected.insert(selected.end(), backup.begin(), backup.end());
            items.swap(selected);
            return true;
        }
        return DoSort(&items, d, err);
    }
};

}  // namespace

// ---- rank service --------------------------------------------------------------------------------
namespace rank {
bool Rank(Engine* e, module::User* user, std::vector<module::ItemPtr>& items, context::RecommendContext* ctx,
          std::string* err) {
    const std::string scene = ctx->GetParameter("scene");
    auto rc = e->config.RankConf.find(scene);
    if (rc == e->config.RankConf.end()) return true;                         // no rank config: no rank
    const recconf::RankConfig& conf = rc->second;
    if (conf.RankAlgoList.empty() && conf.RankScore.empty()) return true;    // :168-171
    const int batch = conf.BatchCount > 0 ? conf.BatchCount : 100;           // :163-166
    std::string uv, verr;
    std::vector<float> user_vec;
    if (e->user_vectors.VectorString(user->Id, &uv, &verr)) user_vec = recall::ParseVectorString(uv);
    // batches × algos; a failing batch keeps its previous scores (:274-276,311)
    for (size_t b0 = 0; b0 < items.size(); b0 += (size_t)batch) {
        const size_t b1 = std::min(items.size(), b0 + (size_t)batch);
        for (const auto& algo : conf.RankAlgoList) {
            algorithm::AlgoData data;
            data.kind = algorithm::AlgoData::kRank;
            data.rank.UserVector = user_vec;
            auto uf = e->user_fields.find(user->Id);
            if (uf != e->user_fields.end()) data.rank.UserFieldIds = uf->second;
            for (size_t i = b0; i < b1; ++i) data.rank.ItemIds.push_back(items[i]->Id);
            algorithm::AlgoResult res;
            std::string aerr;
            if (!e->algorithms.Run(algo, data, &res, &aerr)) continue;       // logged; batch skipped
            for (size_t j = 0; j < res.responses.size() && b0 + j < b1; ++j) {          // rank_service.go:314-335
                const algorithm::AlgoResponse& r = res.responses[j];
                if (r.GetModuleType()) {
                    for (const auto& kv : r.GetScoreMap()) items[b0 + j]->AddAlgoScore(algo + "_" + kv.first, kv.second);
                } else if (r.IsMultiClassify()) {
                    for (const auto& kv : r.GetClassifyMap()) {
                        if (kv.second.size() == 1) items[b0 + j]->AddAlgoScore(algo + "_" + kv.first, kv.second[0]);
                        else for (size_t c = 0; c < kv.second.size(); ++c)
                            items[b0 + j]->AddAlgoScore(algo + "_" + kv.first + "_" + std::to_string(c), kv.second[c]);
                    }
                } else {
                    items[b0 + j]->AddAlgoScore(algo, r.GetScore());
                }
            }
        }
    }
    if (!conf.RankScore.empty()) {                                           // :339-363
        const uint32_t n = (uint32_t)items.size();
        // one expression over every item: variables from the AB parameters first, then the item (ast_parameter_data.go:30-40)
        auto eval = [&](pg_expr* ex, std::vector<double>* out) -> bool {
            const int nv = pg_expr_num_vars(ex);
            const bool antlr = pg_expr_is_antlr(ex) != 0;
            std::vector<double> vars((size_t)nv * n, 0.0);
            std::vector<char> failed(antlr ? n : 0, 0);
            out->assign(n, 0.0);
            for (int v = 0; v < nv; ++v) {
                const std::string name = pg_expr_var_name(ex, v);
                auto ab = ctx->ExperimentParams.find(name);
                if (!antlr) {
                    for (uint32_t i = 0; i < n; ++i) {
                        double x = 0.0;
                        if (ab != ctx->ExperimentParams.end() && ab->second != 0.0) x = ab->second;
                        else items[i]->FloatExprData(name, &x);
                        vars[(size_t)v * n + i] = x;
                    }
                    continue;
                }
                // ASTType "antlr": the data is AstParameterData.ExprData() — the experiment's parameters overlaid by the item's
                // algorithm scores and properties (ast_parameter_data.go:17-28, item.go:213-227; no "current_score" there); a
                // variable the map lacks, or one that is not a number, fails the evaluation: the item scores 0 (ast.go:374-383).
                // maxIndex(${p}) / maxValue(${p}) (antlr_functions.go:34-66,72-91) read the list property p.
                const bool fidx = name.rfind("maxIndex(", 0) == 0, fval = name.rfind("maxValue(", 0) == 0;
                const std::string inner = (fidx || fval) ? name.substr(9, name.size() - 10) : name;
                for (uint32_t i = 0; i < n; ++i) {
                    double x = 0.0;
                    bool ok = false;
                    const module::Item& it = *items[i];
                    auto a = it.algoScores.find(inner);
                    auto p = it.Properties.find(inner);
                    if (fidx || fval) {
                        if (p != it.Properties.end() && p->second.type == json::Value::Array && !p->second.arr.empty()) {
                            size_t best = 0;
                            double bv = ToFloat(p->second.arr[0], 0.0);
                            for (size_t j = 1; j < p->second.arr.size(); ++j) {
                                const double cv = ToFloat(p->second.arr[j], 0.0);
                                if (cv > bv) { bv = cv; best = j; }
                            }
                            x = fidx ? (double)best : bv;
                            ok = true;
                        }
                    } else if (p != it.Properties.end()) {                  // (the item's properties are written last: they win)
                        ok = p->second.type == json::Value::Number;
                        x = p->second.num;
                    } else if (a != it.algoScores.end()) {
                        x = a->second;
                        ok = true;
                    } else if (ab != ctx->ExperimentParams.end()) {
                        x = ab->second;
                        ok = true;
                    }
                    if (!ok) failed[i] = 1;
                    vars[(size_t)v * n + i] = x;
                }
            }
            if (n && pg_expr_eval(e->ctx, ex, nv ? vars.data() : nullptr, n, out->data()) != PG_OK) return false;
            for (uint32_t i = 0; antlr && i < n; ++i)
                if (failed[i]) (*out)[i] = 0.0;
            return true;
        };
        pg_expr* ex = nullptr;
        if (pg_expr_compile_typed(conf.RankScore.c_str(), conf.ASTType.c_str(), &ex) != PG_OK) { if (err) *err = pg_err("pg_expr_compile"); return false; }
        // ScoreRewrite (:296-306,343-353): every source's expression over the item as the algorithms left it, into a map
        // first, then written back (AddAlgoScores); a source whose expression does not compile scores 0
        if (!conf.ScoreRewrite.empty()) {
            std::vector<std::pair<std::string, std::vector<double>>> rewritten;
            for (const auto& rw : conf.ScoreRewrite) {
                std::vector<double> vals(n, 0.0);
                pg_expr* rex = nullptr;
                if (pg_expr_compile_typed(rw.second.c_str(), conf.ASTType.c_str(), &rex) == PG_OK) {
                    const bool ok = eval(rex, &vals);
                    pg_expr_free(rex);
                    if (!ok) { pg_expr_free(ex); if (err) *err = pg_err(("pg_expr_eval (ScoreRewrite[" + rw.first + "])").c_str()); return false; }
                }
                rewritten.emplace_back(rw.first, std::move(vals));
            }
            for (uint32_t i = 0; i < n; ++i) {
                std::map<std::string, double> scores;
                for (const auto& rw : rewritten) scores[rw.first] = rw.second[i];
                items[i]->AddAlgoScores(scores);
            }
        }
        std::vector<double> out;
        const bool ok = eval(ex, &out);
        pg_expr_free(ex);
        if (!ok) { if (err) *err = pg_err("pg_expr_eval"); return false; }
        for (uint32_t i = 0; i < n; ++i) items[i]->Score = out[i];
    }
    return true;
}
}  // namespace rank

// ---- engine --------------------------------------------------------------------------------------
Engine::~Engine() {
    if (ctx) {
        for (auto& kv : co_page) {
            pg_coalescer_destroy(kv.second.first);
            pg_expr_free(kv.second.second);
        }
        for (auto& kv : co_scene) pg_coalescer_destroy(kv.second);
        for (auto& kv : co_online) pg_coalescer_destroy(kv.second);
        DropViewsLocked();
        if (model) pg_model_destroy(ctx, model);
        for (auto& kv : named_models) pg_model_destroy(ctx, kv.second);
        if (fm2t) pg_model_destroy(ctx, fm2t);
        if (feats) pg_features_destroy(ctx, feats);
        if (staging_feats) pg_features_destroy(ctx, staging_feats);
        if (item_emb) pg_table_destroy(ctx, item_emb);
        if (staging) pg_table_destroy(ctx, staging);
        if (table) pg_table_destroy(ctx, table);
        pg_shutdown(ctx);
    }
}

// One coalescer per recall depth serves every per-request plug-in call of the scene: vector / i2i recalls, both rank
// algorithms (the DNN as algorithm 0, FM + two-tower behind it) and DPPSort.  k = 0: any (rank / sort calls do not
// depend on the recall depth).  At most kMaxSceneCoalescers depths are kept (each holds slots and a sibling context's
// scratch); a request with yet another K is served by the direct call.
pg_coalescer* Engine::SceneCoalescer(uint32_t k, std::string* err) {
    std::lock_guard<std::mutex> g(co_mu);
    if (k == 0 && !co_scene.empty()) return co_scene.begin()->second;
    if (k == 0) k = 5000;                          // sizes a rank batch: as many candidates as 256 requests x 5000
    auto it = co_scene.find(k);
    if (it != co_scene.end()) return it->second;
    if (co_scene.size() >= kMaxSceneCoalescers) {
        if (err) *err = "coalescer: too many distinct recall depths";
        return nullptr;
    }
    pg_scene_config sc;
    memset(&sc, 0, sizeof sc);
    sc.base.k = k;
    sc.base.max_wait_us = coalesce_wait_us;
    sc.base.depth = coalesce_depth;
    sc.base.max_rank_items = 16384;
    sc.base.timeout_us = coalesce_timeout_us;
    pg_rank_algo algos[2];
    memset(algos, 0, sizeof algos);
    std::vector<int32_t> cols;
    if (model) {
        algos[sc.n_algos].model = model;
        algos[sc.n_algos].name = "dnn";
        sc.n_algos++;
    }
    if (fm2t && feats && !fm2t_columns.empty()) {
        bool ok = true;
        for (const auto& cname : fm2t_columns) {
            const int idx = pg_features_column_index(feats, cname.c_str());
            ok = ok && idx >= 0;
            cols.push_back(idx);
        }
        if (ok) {
            algos[sc.n_algos].model = fm2t;
            algos[sc.n_algos].name = "fm2t";
            algos[sc.n_algos].features = feats;
            algos[sc.n_algos].item_field_cols = cols.data();
            sc.n_algos++;
        }
    }
    sc.algos = algos;
    pg_coalescer* c = nullptr;
    if (pg_coalescer_create_scene(ctx, table, &sc, &c) != PG_OK) {
        if (err) *err = std::string("pg_coalescer_create_scene: ") + pg_last_error();
        return nullptr;
    }
    co_scene[k] = c;
    return c;
}

// the OnlineVectorRecall's coalescer: user features → the vector model's user tower → top-k of the item-embedding table
pg_coalescer* Engine::OnlineCoalescer(uint32_t k, std::string* err) {
    std::lock_guard<std::mutex> g(co_mu);
    auto it = co_online.find(k);
    if (it != co_online.end()) return it->second;
    if (co_online.size() >= kMaxSceneCoalescers) {
        if (err) *err = "coalescer: too many distinct recall depths";
        return nullptr;
    }
    pg_scene_config sc;
    memset(&sc, 0, sizeof sc);
    sc.base.k = k;
    sc.base.max_wait_us = coalesce_wait_us;
    sc.base.depth = coalesce_depth;
    sc.base.timeout_us = coalesce_timeout_us;
    sc.query_model = fm2t;
    pg_coalescer* c = nullptr;
    if (pg_coalescer_create_scene(ctx, item_emb, &sc, &c) != PG_OK) {
        if (err) *err = std::string("pg_coalescer_create_scene: ") + pg_last_error();
        return nullptr;
    }
    co_online[k] = c;
    return c;
}

// configuration-time changes (a model or a feature column is replaced) retire the coalescers that hold the old objects;
// the next call builds new ones.  Not to be called while requests are in flight.
void Engine::DropCoalescers() {
    std::lock_guard<std::mutex> g(co_mu);
    for (auto& kv : co_page) {
        pg_coalescer_destroy(kv.second.first);
        pg_expr_free(kv.second.second);
    }
    co_page.clear();
    for (auto& kv : co_scene) pg_coalescer_destroy(kv.second);
    co_scene.clear();
    for (auto& kv : co_online) pg_coalescer_destroy(kv.second);
    co_online.clear();
    DropViewsLocked();
}

void Engine::DropViewsLocked() {
    for (auto& kv : co_views) {
        if (kv.second.co) pg_coalescer_destroy(kv.second.co);
        if (kv.second.view) pg_table_destroy(ctx, kv.second.view);
    }
    co_views.clear();
}

// The view of a filtered Hologres recall for the table's current generation.  The constant of the clause was fixed when the
// recall was built (hologres_vector_recall.go:56-61), so the admitted set only changes with the table (ingest.cpp: a commit is
// exclusive against every request, so no request sees two generations) or with the column (ph_engine_set_feature_column drops
// the views).  nullptr: no feature column of that name (the SQL would fail).
const Engine::FilterView* Engine::ViewFor(const recconf::RecallConfig& conf, uint32_t k, const pg_table* base) {
    if (!base) base = table;
    std::lock_guard<std::mutex> g(co_mu);
    const int col = feats ? pg_features_column_index(feats, conf.WhereColumn.c_str()) : -1;
    if (col < 0) return nullptr;
    FilterView& fv = co_views[conf.Name];
    const uint64_t gen = generation.load();
    if (fv.generation == gen) return &fv;
    if (fv.co) pg_coalescer_destroy(fv.co);
    if (fv.view) pg_table_destroy(ctx, fv.view);
    fv = FilterView();
    fv.generation = gen;
    const int rc = pg_table_view_create(ctx, base, feats, col, conf.WhereOp, conf.WhereValue, &fv.view);
    if (rc == PG_ERR_EMPTY) { fv.empty = true; return &fv; }             // no row passes: every answer is empty
    // (PG_ERR_INVALID is a misconfiguration — a feature store shorter than the table, a bad operator: the per-call path below
    // reports it on every request instead of serving empty pages for the whole generation)
    if (rc != PG_OK) { fv.failed = true; fv.view = nullptr; return &fv; }
    if (coalesce) {
        pg_scene_config sc;
        memset(&sc, 0, sizeof sc);
        sc.base.k = k;
        sc.base.max_wait_us = coalesce_wait_us;
        sc.base.depth = coalesce_depth;
        sc.base.timeout_us = coalesce_timeout_us;
        if (pg_coalescer_create_scene(ctx, fv.view, &sc, &fv.co) != PG_OK) fv.co = nullptr;    // direct calls on the view then
    }
    return &fv;
}

pg_coalescer* Engine::PageCoalescer(const recconf::RecallConfig& conf, std::string* err) {
    std::lock_guard<std::mutex> g(co_mu);
    auto it = co_page.find(conf.Name);
    if (it != co_page.end()) return it->second.first;
    if (!model) {
        if (err) *err = "page recall: no rank model loaded";
        return nullptr;
    }
    pg_expr* ex = nullptr;
    if (pg_expr_compile(conf.RankScore.c_str(), &ex) != PG_OK) {
        if (err) *err = std::string("pg_expr_compile: ") + pg_last_error();
        return nullptr;
    }
    if (!conf.ScoreRewrite.empty()) {
        // the scene's ScoreRewrite rides on the RankScore: the device evaluates every source from the un-rewritten scores
        // in front of it (a source that does not compile scores 0, as in rank_service.go:299-303,349-351)
        std::vector<const char*> src;
        std::vector<pg_expr*> rex;
        for (const auto& rw : conf.ScoreRewrite) {
            pg_expr* r = nullptr;
            if (pg_expr_compile(rw.second.c_str(), &r) != PG_OK) r = nullptr;
            src.push_back(rw.first.c_str());
            rex.push_back(r);
        }
        const int rc = pg_expr_set_score_rewrites(ex, (uint32_t)src.size(), src.data(), rex.data());
        for (pg_expr* r : rex) pg_expr_free(r);
        if (rc != PG_OK) {
            if (err) *err = std::string("pg_expr_set_score_rewrites: ") + pg_last_error();
            pg_expr_free(ex);
            return nullptr;
        }
    }
    pg_coalescer_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.k = (uint32_t)conf.RecallCount;
    cfg.max_wait_us = coalesce_wait_us;
    cfg.depth = coalesce_depth;
    cfg.max_top_n = (uint32_t)std::min(conf.RecallCount, 1000);
    cfg.timeout_us = coalesce_timeout_us;
    pg_coalescer* c = nullptr;
    if (pg_coalescer_create(ctx, table, model, ex, conf.RankVar.c_str(), &cfg, &c) != PG_OK) {
        if (err) *err = std::string("pg_coalescer_create: ") + pg_last_error();
        pg_expr_free(ex);
        return nullptr;
    }
    co_page[conf.Name] = std::make_pair(c, ex);
    return c;
}

Engine* Engine::Create(const std::string& config_json, std::string* err) {
    std::unique_ptr<Engine> e(new Engine());
    if (!recconf::RecommendConfig::Parse(config_json, &e->config, err)) return nullptr;
    // What the reference's own loaders would do with RecallConfs / SortConfs comes first (they run in runBeforeStart,
    // before any start hook): a config that panics pairec must not start here either.
    for (const auto& r : e->config.RecallConfs) {
        const recall::LoadOutcome o = recall::CheckRecallConf(r);
        if (o.kind == recall::LoadOutcome::kPanic) { if (err) *err = "panic: " + o.message; return nullptr; }
        if (o.kind == recall::LoadOutcome::kUnavailable) { if (err) *err = o.message; return nullptr; }
    }
    // RankConf.ASTType = "antlr" selects another expression language in the reference (GetExpASTWithType /
    // ExprASTResultWithType, utils/ast/ast.go:338-389: the valuate evaluator with AntlrFunctions — maxIndex, maxValue, …).
    // This engine evaluates the default grammar only; evaluating an "antlr" RankScore with it would silently change scores.
    // Round 5: the subset of that language the reference's tests pin is compiled to the device program
    // (pg_expr_compile_typed); a RankScore or ScoreRewrite expression outside it still stops the load, named.
    for (const auto& kv : e->config.RankConf)
        if (kv.second.ASTType == "antlr") {
            std::vector<std::pair<std::string, std::string>> exprs;
            exprs.emplace_back("RankScore", kv.second.RankScore);
            for (const auto& rw : kv.second.ScoreRewrite) exprs.emplace_back("ScoreRewrite[" + rw.first + "]", rw.second);
            for (const auto& ex : exprs) {
                pg_expr* probe = nullptr;
                if (pg_expr_compile_typed(ex.second.c_str(), "antlr", &probe) != PG_OK) {
                    if (err) *err = "RankConf[" + kv.first + "]." + ex.first + " with ASTType \"antlr\": " + pg_last_error();
                    return nullptr;
                }
                pg_expr_free(probe);
            }
        }
    // FeatureService.LoadFeatureConf (feature_service.go:33-58) / UserFeatureService: one Feature per FeatureLoadConfig.  An unknown
    // FeatureType is the reference's panic (op.go:32); an expression normalizer outside the stated subset stops the load, named.
    {
        auto build = [&](const std::map<std::string, recconf::SceneFeatureConfig>& src, const char* key,
                         std::map<std::string, std::vector<std::shared_ptr<feature::Feature>>>* dst) {
            for (const auto& sc : src)
                for (const auto& lc : sc.second.FeatureLoadConfs) {
                    auto f = std::make_shared<feature::Feature>();
                    std::string ferr;
                    if (!f->LoadWithConfig(lc.Features, &ferr)) {
                        if (err) *err = std::string(key) + "[" + sc.first + "]: " + ferr;
                        return false;
                    }
                    (*dst)[sc.first].push_back(std::move(f));
                }
            return true;
        };
        if (!build(e->config.FeatureConfs, "FeatureConfs", &e->sceneFeatures)) return nullptr;
        if (!build(e->config.UserFeatureConfs, "UserFeatureConfs", &e->sceneUserFeatures)) return nullptr;
    }
    for (const auto& sc : e->config.SortConfs) {
        // RegisterSortWithConfig (sort/sort.go:162-200): DPPSort / SSDSort open their Hologres datasource first
        // (NewDPPSort, dpp_sort.go:60-64; NewSSDSort, ssd_sort.go:52-56) and panic without it; an unknown SortType
        // leaves s nil → panic("Sort is nil, name:…")
        if (sc.SortType == "DPPSort" || sc.SortType == "SSDSort") {
            if (err) *err = "panic: Postgres not found, name:" + sc.HologresName + " (SortConfs entry \"" + sc.Name + "\": " + sc.SortType +
                            " opens its Hologres datasource in the constructor; GPU diversity sorts are declared in UserDefineConfs.pairec_gpu.Sorts)";
            return nullptr;
        }
        static const char* host_sorts[] = {"AlgoScoreSort", "MultiRecallMixSort", "BoostScoreSort", "DiversityRuleSort", "TrafficControlSort",
                                           "BoostScoreByWeight", "DistinctIdSort", "CustomFieldSort", "ConditionSort"};
        bool known = false;
        for (const char* r : host_sorts) known |= sc.SortType == r;
        if (!known) { if (err) *err = "panic: Sort is nil, name:" + sc.Name; return nullptr; }
    }
    // GPU plugin settings live in UserDefineConfs["pairec_gpu"]: they cannot be declared in RecallConfs / SortConfs
    const json::Value& g = e->config.UserDefineConfs.at("pairec_gpu");
    if (g.type != json::Value::Object) { if (err) *err = "UserDefineConfs.pairec_gpu missing"; return nullptr; }
    const json::Value& tb = g.at("Table");
    e->table_rows = (uint64_t)tb.n("Rows");
    e->dim = (uint32_t)tb.n("Dim", 128);
    e->id_prefix = tb.s("IdPrefix", "item_");
    if (pg_init((int)g.n("Device", 0), nullptr, &e->ctx) != PG_OK) { if (err) *err = pg_err("pg_init"); return nullptr; }
    if (pg_table_create(e->ctx, e->table_rows, e->dim, 0, &e->table) != PG_OK) { if (err) *err = pg_err("pg_table_create"); return nullptr; }
    const std::string table_path = tb.s("Path", "");
    if (!table_path.empty()) {
        // (item_id, embedding) rows from a file: the loader streams them in chunks and commits (ingest.cpp); the same
        // call replaces the table later while the engine serves (ph_engine_ingest_file)
        if (!e->IngestFile(table_path, err)) return nullptr;
    } else if (pg_table_fill_synthetic(e->ctx, e->table, (uint64_t)tb.n("SyntheticSeed", 0x5EED0001), 1) != PG_OK) {
        if (err) *err = pg_err("pg_table_fill_synthetic");
        return nullptr;
    }
    const json::Value& co = g.at("Coalesce");
    if (co.type == json::Value::Object) {
        e->coalesce = true;
        e->coalesce_wait_us = (uint32_t)co.n("MaxWaitUs", 0);
        e->coalesce_depth = (uint32_t)co.n("Depth", 0);
        e->coalesce_timeout_us = (uint32_t)co.n("TimeoutUs", 0);       // the EAS client's per-call timeout (eas/client.go:53-58)
    }
    const json::Value& ov = g.at("OnlineVector");          // the vector model's item side: item-tower outputs as a table
    if (ov.type == json::Value::Object) {
        e->item_emb_rows = (uint64_t)ov.n("Rows", (long long)e->table_rows);
        e->item_emb_dim = (uint32_t)ov.n("Dim", 64);
        if (pg_table_create(e->ctx, e->item_emb_rows, (uint32_t)ov.n("Dim", 64), 0, &e->item_emb) != PG_OK ||
            pg_table_fill_synthetic(e->ctx, e->item_emb, (uint64_t)ov.n("SyntheticSeed", 0x5EED0077), 1) != PG_OK) {
            if (err) *err = pg_err("online vector item table");
            return nullptr;
        }
    }
    // built-in sorts (sort.go init(): "ItemRankScore" is registered by default)
    e->sorts.RegisterSort("ItemRankScore", std::make_shared<GpuItemRankScoreSort>(e.get()), nullptr);
    e->sorts.RegisterSort("ItemScore", std::make_shared<GpuItemScoreSort>(e.get()), nullptr);
    for (const auto& d : e->config.DPPConf) e->sorts.RegisterSort(d.Name, std::make_shared<GpuDPPSort>(e.get(), d), nullptr);
    // SortConfs (validated above): AlgoScoreSort and CustomFieldSort are host logic over a GPU sort; the other rule-based sorts
    // are outside this engine's scope and are skipped
    for (const auto& sc : e->config.SortConfs) {
        if (sc.SortType == "AlgoScoreSort") e->sorts.RegisterSortWithConfig(sc.Name, std::make_shared<GpuAlgoScoreSort>(e.get(), sc));
        else if (sc.SortType == "CustomFieldSort") e->sorts.RegisterSortWithConfig(sc.Name, std::make_shared<GpuCustomFieldSort>(e.get(), sc));
    }
    // the GPU sorts of a pairec process: registered by name in the start hook (sort.RegisterSort — first registration wins)
    for (const auto& sc : e->config.GpuSorts) {
        if (sc.SortType == "DPPSort") e->sorts.RegisterSort(sc.Name, std::make_shared<GpuDPPSort>(e.get(), sc.DPPConf), nullptr);
        else if (sc.SortType == "SSDSort") e->sorts.RegisterSort(sc.Name, std::make_shared<GpuSSDSort>(e.get(), sc.SSDConf), nullptr);
        else if (sc.SortType == "AlgoScoreSort") e->sorts.RegisterSort(sc.Name, std::make_shared<GpuAlgoScoreSort>(e.get(), sc), nullptr);
        else if (sc.SortType == "CustomFieldSort") e->sorts.RegisterSort(sc.Name, std::make_shared<GpuCustomFieldSort>(e.get(), sc), nullptr);
        else if (sc.SortType == "ItemRankScore") e->sorts.RegisterSort(sc.Name, std::make_shared<GpuItemRankScoreSort>(e.get()), nullptr);
        else if (sc.SortType == "ItemScore") e->sorts.RegisterSort(sc.Name, std::make_shared<GpuItemScoreSort>(e.get()), nullptr);
        else { if (err) *err = "pairec_gpu.Sorts: unknown SortType " + sc.SortType; return nullptr; }
    }
    // algorithms by name (the shim's start hook does the same with algorithm.RegisterAlgorithm)
    bool first_dnn = true, first_fm = true;
    for (const auto& a : g.at("Algorithms").arr) {
        const std::string name = a.s("Name"), kind = a.s("Kind");
        if (a.has("Precision")) {
            // the rank model's matrix layers: "f32" (bit-defined), "bf16" (fastest), "bf16x3" (split bf16: the fp32 scores of
            // the reference's model servers — eas/easyrec_response.go:479-483 — at matrix-pipe speed)
            const std::string pr = a.s("Precision");
            const int pv = pr == "f32" ? PG_PREC_F32 : pr == "bf16" ? PG_PREC_BF16 : pr == "bf16x3" ? PG_PREC_BF16X3 : -1;
            if (pv < 0) {
                if (err) *err = "pairec_gpu.Algorithms: " + name + ": Precision \"" + pr + "\" (f32, bf16 or bf16x3)";
                return nullptr;
            }
            e->algo_precision[name] = pv;
            if (kind == "dnn3" && first_dnn) e->default_dnn_precision = pv;
            if (kind == "fm2t" && first_fm) e->default_fm2t_precision = pv;
        }
        first_dnn = first_dnn && kind != "dnn3";
        first_fm = first_fm && kind != "fm2t";
        if (kind == "faiss") e->algorithms.RegisterAlgorithm(name, std::make_shared<GpuFaissAlgorithm>(e.get()));
        else if (kind == "dnn3") {
            std::vector<std::string> outs;
            for (const auto& o : a.at("Outputs").arr) if (o.type == json::Value::String) outs.push_back(o.str);
            e->algorithms.RegisterAlgorithm(name, std::make_shared<GpuDnnAlgorithm>(e.get(), name, outs));
        }
        else if (kind == "online_vector") e->algorithms.RegisterAlgorithm(name, std::make_shared<GpuOnlineVectorAlgorithm>(e.get()));
        else if (kind == "fm2t") {
            std::vector<std::string> cols;
            for (const auto& c : a.at("ItemFieldColumns").arr) if (c.type == json::Value::String) cols.push_back(c.str);
            if (e->fm2t_columns.empty()) e->fm2t_columns = cols;       // the scene coalescer's FM algorithm uses the first one's columns
            e->algorithms.RegisterAlgorithm(name, std::make_shared<GpuFm2tAlgorithm>(e.get(), cols));
        }
    }
    // recall.Load over RecallConfs, with the reference's outcomes (service/recall/recall.go:47-107)
    for (const auto& r : e->config.RecallConfs) {
        if (r.RecallType == "MockRecall") e->recalls.RegisterRecall(r.Name, std::make_shared<MockRecall>(r));
        else if (r.RecallType == "OnlineVectorRecall") e->recalls.RegisterRecall(r.Name, std::make_shared<GpuOnlineVectorRecall>(e.get(), r));
    }
    // the GPU recalls: recall.RegisterRecall(name, impl) in the start hook (overwrites)
    for (const auto& r : e->config.GpuRecalls) {
        if (r.Kind == "vector") e->recalls.RegisterRecall(r.Name, std::make_shared<GpuVectorRecall>(e.get(), r));
        else if (r.Kind == "i2i") e->recalls.RegisterRecall(r.Name, std::make_shared<GpuI2IVectorRecall>(e.get(), r));
        else if (r.Kind == "hologres_v2") e->recalls.RegisterRecall(r.Name, std::make_shared<GpuHologresVectorRecall>(e.get(), r, true));
        else if (r.Kind == "hologres") e->recalls.RegisterRecall(r.Name, std::make_shared<GpuHologresVectorRecall>(e.get(), r, false));
        else if (r.Kind == "online_hologres") e->recalls.RegisterRecall(r.Name, std::make_shared<GpuOnlineHologresVectorRecall>(e.get(), r));
        else if (r.Kind == "page") {
            if (r.RankScore.empty() || r.RankVar.empty() || r.RecallCount <= 0) {
                if (err) *err = "pairec_gpu.Recalls: Kind \"page\" needs RecallCount, RankScore and RankVar";
                return nullptr;
            }
            e->recalls.RegisterRecall(r.Name, std::make_shared<GpuPageRecall>(e.get(), r));
        }
        else { if (err) *err = "pairec_gpu.Recalls: unknown Kind " + r.Kind; return nullptr; }
    }
    return e.release();
}

bool Engine::Recommend(const std::string& uid, int size, const std::string& scene,
                       std::vector<module::ItemPtr>* out, std::string* err) {
    return Recommend(uid, size, scene, json::Value(), out, err);
}

bool Engine::Recommend(const std::string& uid, int size, const std::string& scene, const json::Value& experiment_params,
                       std::vector<module::ItemPtr>* out, std::string* err) {
    VersionLock::Read generation_guard(version);     // one generation of rows and ids for the whole request
    module::User user(uid);
    context::RecommendContext ctx;
    ctx.Size = size;
    ctx.Param["scene"] = json::Value::Str(scene);
    // request parameters other than the scene ride in the experiment object under "_param" (the driver API has no
    // separate argument for them): "item_id" is what I2IVectorRecall reads with context.GetParameter
    for (const auto& kv : experiment_params.at("_param").obj) ctx.Param[kv.first] = kv.second;
    ctx.ExperimentParamsJson = experiment_params;
    // the request's user features ("features" of POST /api/recommend → User.Properties, web/recommend_controller.go) ride under
    // "_user_features"; UserFeatureService.LoadUserFeatures (user_recommend.go:64) transforms them before the recalls
    for (const auto& kv : experiment_params.at("_user_features").obj) user.Properties[kv.first] = kv.second;
    {
        auto uf = sceneUserFeatures.find(scene);
        if (uf != sceneUserFeatures.end()) {
            std::vector<module::ItemPtr> none;
            for (const auto& f : uf->second) f->LoadFeatures(&user, none, &ctx);
        }
    }
    // RecallService.GetItems (service/recall.go:53-153): scene → category → recall names, concatenated
    std::vector<module::ItemPtr> items;
    auto sc = config.SceneRecallNames.find(scene);
    if (sc != config.SceneRecallNames.end())
        for (const auto& cat : sc->second)
            for (const auto& name : cat.second) {
                std::string rerr;
                auto r = recalls.GetRecall(name, &rerr);
                if (!r) continue;
                // AB experiment: interface assertion + "recall.<name>" params object (service/recall.go:95-105)
                if (ctx.HasExperiment())
                    if (auto* cr = dynamic_cast<recall::ICloneRecall*>(r.get())) {
                        const json::Value& rc = ctx.ExperimentParamsJson.at("recall." + cr->GetRecallName());
                        if (rc.type == json::Value::Object)
                            if (auto cloned = cr->CloneWithConfig(rc)) r = cloned;
                    }
                auto got = r->GetCandidateItems(&user, &ctx);
                items.insert(items.end(), got.begin(), got.end());
            }
    items = filter::UniqueFilter(items);
    {   // FeatureService.LoadFeatures (user_recommend.go:129; feature_service.go:77-131): "features.scene.name" of the experiment first
        std::string fscene = ctx.HasExperiment() ? ctx.ExperimentParamsJson.s("features.scene.name") : "";
        if (fscene.empty()) fscene = scene;
        auto sf = sceneFeatures.find(fscene);
        if (sf != sceneFeatures.end())
            for (const auto& f : sf->second) f->LoadFeatures(&user, items, &ctx);
    }
    if (!rank::Rank(this, &user, items, &ctx, err)) return false;
    // SortService.Sort (sort/sort.go:65-125): SortNames[scene] else the default ItemRankScore
    std::vector<std::string> names;
    auto sn = config.SortNames.find(scene);
    if (sn != config.SortNames.end()) names = sn->second;
    if (names.empty()) names.push_back("ItemRankScore");
    sort::SortData sd;
    sd.Data = items;
    sd.Context = &ctx;
    sd.User = &user;
    for (const auto& n : names) {
        auto s = sorts.Get(n);
        if (!s) continue;
        // AB experiment: ICloneSort + "sort.<name>" params object (sort/sort.go:110-121)
        if (ctx.HasExperiment())
            if (auto* cs = dynamic_cast<sort::ICloneSort*>(s.get())) {
                const json::Value& sc2 = ctx.ExperimentParamsJson.at("sort." + cs->GetSortName());
                if (sc2.type == json::Value::Object)
                    if (auto cloned = cs->CloneWithConfig(sc2)) s = cloned;
            }
        std::string serr;
        s->Sort(&sd, &serr);                                          // error ignored (sort.go:123)
    }
    if ((int)sd.Data.size() > size) sd.Data.resize((size_t)size);     // items[:size]
    *out = sd.Data;
    return true;
}

// ---- EasyrecAlgoDataGenerator ------------------------------------------------------------------------
namespace rank {
static void value_json(const json::Value& v, std::string* o) {
    switch (v.type) {
        case json::Value::String: json::Escape(v.str, o); break;
        case json::Value::Number: *o += v.is_int ? std::to_string(v.i) : json::NumToString(v.num); break;
        case json::Value::Bool: *o += v.b ? "true" : "false"; break;
        default: *o += "null";
    }
}
json::Value EasyrecAlgoDataGenerator::Feature::Default() const {       // feature.defaultValue, :154-171
    json::Value v;
    if (type == json::Value::Number) {
        v.type = json::Value::Number;
        v.is_int = is_int;
        v.num = 0.0;
        v.i = 0;
    } else {
        v.type = json::Value::String;                                  // string and everything else: ""
    }
    return v;
}
EasyrecAlgoDataGenerator::EasyrecAlgoDataGenerator(const std::vector<std::string>& contextFeatures) {
    // parseFeature starts true (:187): the schema is the configured list, every column typed string
    for (const auto& n : contextFeatures) itemFeatures_.push_back(Feature{n, json::Value::String, false});
}
void EasyrecAlgoDataGenerator::SetItemFeatures(const std::vector<std::string>& in) {
    if (!in.empty()) {
        hasInputMap_ = true;
        if (in[0] != "*") {
            parseInputItemFeature_ = true;
            for (const auto& n : in) inputItemFeatures_.push_back(Feature{n, json::Value::String, false});
        }
    } else {
        parseInputItemFeature_ = true;
    }
}
void EasyrecAlgoDataGenerator::AddFeatures(const module::ItemPtr& item,
                                           const std::map<std::string, json::Value>& itemFeatures,
                                           const std::map<std::string, json::Value>& userFeatures) {
    if (item) requestItem_.push_back(item);
    if (!parseFeature_) {                       // (never taken after the constructor above; kept for fidelity)
        for (const auto& kv : itemFeatures) itemFeatures_.push_back(Feature{kv.first, kv.second.type, kv.second.is_int});
        userFeatures_ = userFeatures;
        parseFeature_ = true;
    }
    if (!parseInputItemFeature_) {              // "*": every feature of the first item that is not a context feature
        for (const auto& kv : itemFeatures) {
            bool ctxf = false;
            for (const auto& f : itemFeatures_) ctxf |= f.name == kv.first;
            if (!ctxf) inputItemFeatures_.push_back(Feature{kv.first, kv.second.type, kv.second.is_int});
        }
        parseInputItemFeature_ = true;
    }
    if (userFeatures_.empty()) userFeatures_ = userFeatures;
    for (const auto& f : itemFeatures_) {
        auto it = itemFeatures.find(f.name);
        contextFeatures_[f.name].push_back(it != itemFeatures.end() ? it->second : f.Default());
    }
    if (hasInputMap_)
        for (const auto& f : inputItemFeatures_) {
            auto it = itemFeatures.find(f.name);
            inputItemFeatureMap_[f.name].push_back(it != itemFeatures.end() ? it->second : f.Default());
        }
}
std::string EasyrecAlgoDataGenerator::GeneratorAlgoData() {
    std::string o = "{\"user_features\":{";
    bool first = true;
    for (const auto& kv : userFeatures_) {
        if (!first) o += ",";
        first = false;
        json::Escape(kv.first, &o);
        o += ":";
        value_json(kv.second, &o);
    }
    o += "},\"item_ids\":[";
    for (size_t i = 0; i < requestItem_.size(); ++i) {
        if (i) o += ",";
        json::Escape(requestItem_[i]->Id, &o);
    }
    o += "]";
    auto lists = [&](const char* key, std::map<std::string, std::vector<json::Value>>& m) {
        o += std::string(",\"") + key + "\":{";
        bool f2 = true;
        for (auto& kv : m) {
            if (!f2) o += ",";
            f2 = false;
            json::Escape(kv.first, &o);
            o += ":[";
            for (size_t i = 0; i < kv.second.size(); ++i) {
                if (i) o += ",";
                value_json(kv.second[i], &o);
            }
            o += "]";
            kv.second.clear();                   // g.contextFeatures[k] = g.contextFeatures[k][:0]
        }
        o += "}";
    };
    lists("context_features", contextFeatures_);
    lists("item_features", inputItemFeatureMap_);
    o += "}";
    requestItem_.clear();
    return o;
}
}  // namespace rank

}  // namespace pairec

// ---- C driver API for the tests ---------------------------------------------------------------------
using namespace pairec;
static thread_local std::string g_ph_err;
static thread_local std::string g_ph_out;

extern "C" {

const char* ph_last_error(void) { return g_ph_err.c_str(); }

void* ph_engine_create(const char* config_json) {
    std::string err;
    Engine* e = Engine::Create(config_json ? config_json : "", &err);
    if (!e) g_ph_err = err;
    return e;
}
void ph_engine_destroy(void* h) { delete (Engine*)h; }

// load the DNN3 rank model (blob format of pg_model_load) into the engine
int ph_engine_load_dnn3(void* h, int prec, const char* blob, size_t len) {
    Engine* e = (Engine*)h;
    if (!e || !blob) return -1;
    VersionLock::Write w(e->version);                  // no request is inside the coalescers / holds the model this replaces
    e->DropCoalescers();                               // they hold the old model
    if (e->model) { pg_model_destroy(e->ctx, e->model); e->model = nullptr; }
    if (prec < 0) prec = e->default_dnn_precision;     // pairec_gpu.Algorithms[].Precision of the scene's DNN
    const int rc = pg_model_load(e->ctx, PG_MODEL_DNN3, (pg_prec)prec, blob, len, &e->model);
    if (rc != PG_OK) g_ph_err = pg_last_error();
    return rc;
}

// the multi-output model of rank algorithm `algo` (blob format PG_MODEL_DNN3_MULTI: n_out heads on one trunk); its
// outputs are named by the algorithm's "Outputs" list, in order
int ph_engine_load_dnn3_multi(void* h, const char* algo, int prec, const char* blob, size_t len) {
    Engine* e = (Engine*)h;
    if (!e || !blob || !algo) return -1;
    VersionLock::Write w(e->version);                  // no request in flight holds the model this replaces
    pg_model* m = nullptr;
    if (prec < 0) prec = e->PrecisionOf(algo, e->default_dnn_precision);
    const int rc = pg_model_load(e->ctx, PG_MODEL_DNN3_MULTI, (pg_prec)prec, blob, len, &m);
    if (rc != PG_OK) { g_ph_err = pg_last_error(); return rc; }
    auto it = e->named_models.find(algo);
    if (it != e->named_models.end()) pg_model_destroy(e->ctx, it->second);
    e->named_models[algo] = m;
    return 0;
}

// one DNN3 model per output of a multi-output rank algorithm: key "<algo>/<output>"
int ph_engine_load_dnn3_named(void* h, const char* key, int prec, const char* blob, size_t len) {
    Engine* e = (Engine*)h;
    if (!e || !blob || !key) return -1;
    VersionLock::Write w(e->version);
    pg_model* m = nullptr;
    if (prec < 0) prec = e->PrecisionOf(std::string(key).substr(0, std::string(key).find('/')), e->default_dnn_precision);
    const int rc = pg_model_load(e->ctx, PG_MODEL_DNN3, (pg_prec)prec, blob, len, &m);
    if (rc != PG_OK) { g_ph_err = pg_last_error(); return rc; }
    auto it = e->named_models.find(key);
    if (it != e->named_models.end()) pg_model_destroy(e->ctx, it->second);
    e->named_models[key] = m;
    return 0;
}

// the vector model of the online recall (blob format of pg_model_load, PG_MODEL_FM_TWOTOWER)
int ph_engine_load_fm2t(void* h, int prec, const char* blob, size_t len) {
    Engine* e = (Engine*)h;
    if (!e || !blob) return -1;
    VersionLock::Write w(e->version);
    e->DropCoalescers();                               // they hold the old model
    if (e->fm2t) { pg_model_destroy(e->ctx, e->fm2t); e->fm2t = nullptr; }
    if (prec < 0) prec = e->default_fm2t_precision;
    const int rc = pg_model_load(e->ctx, PG_MODEL_FM_TWOTOWER, (pg_prec)prec, blob, len, &e->fm2t);
    if (rc != PG_OK) g_ph_err = pg_last_error();
    else if (len >= 16) {
        uint32_t hdr[4];
        memcpy(hdr, blob, sizeof hdr);                  // u32 n_user_fields, n_item_fields, k, d_user (include/pairec_gpu.h)
        e->fm2t_nuf = hdr[0];
        e->fm2t_d_user = hdr[3];
    }
    return rc;
}

// one int32 item feature column (dictionary-encoded categorical feature), keyed by item row
int ph_engine_set_feature_column(void* h, const char* name, const int32_t* values, uint64_t n) {
    Engine* e = (Engine*)h;
    if (!e || !name || !values || n != e->table_rows) { g_ph_err = "feature column: bad argument (one value per table row)"; return -1; }
    // exclusive against every request (they hold the version lock shared from their first plug-in call to their last label
    // lookup): the coalescers and views dropped here are not in use, and no request sees the column half-way.  A column
    // that belongs to a NEW generation of the table goes through ph_engine_ingest_feature_column instead.
    VersionLock::Write w(e->version);
    e->DropCoalescers();
    if (!e->feats && pg_features_create(e->ctx, e->table_rows, &e->feats) != PG_OK) { g_ph_err = pg_last_error(); return -1; }
    const int rc = pg_features_set_column(e->ctx, e->feats, name, PG_F_I32, values, 0.0);
    if (rc != PG_OK) g_ph_err = pg_last_error();
    return rc;
}

int ph_set_user_fields(void* h, const char* uid, const int32_t* ids, int n) {
    if (!h || !uid || !ids || n < 0) return -1;
    ((Engine*)h)->user_fields[uid] = std::vector<int32_t>(ids, ids + n);
    return 0;
}

int ph_set_user_vector(void* h, const char* uid, const char* vec) {
    if (!h || !uid || !vec) return -1;
    ((Engine*)h)->user_vectors.vectors[uid] = vec;
    return 0;
}

// → JSON {"items":[{"item_id":..,"score":..,"retrieve_id":..,"algo_scores":{..}}]}
static const char* items_to_json(const std::vector<module::ItemPtr>& items) {
    std::string& o = g_ph_out;
    o = "{\"items\":[";
    for (size_t i = 0; i < items.size(); ++i) {
        if (i) o += ",";
        o += "{\"item_id\":";
        json::Escape(items[i]->Id, &o);
        o += ",\"score\":" + json::NumToString(items[i]->Score) + ",\"retrieve_id\":";
        json::Escape(items[i]->RetrieveId, &o);
        o += ",\"item_type\":";
        json::Escape(items[i]->ItemType, &o);
        o += ",\"algo_scores\":{";
        bool first = true;
        for (const auto& kv : items[i]->algoScores) {
            if (!first) o += ",";
            first = false;
            json::Escape(kv.first, &o);
            o += ":" + json::NumToString(kv.second);
        }
        o += "}}";
    }
    o += "]}";
    return o.c_str();
}

// one registered sort over caller-made items: [{"id":..,"score":x,"properties":{..},"algo_scores":{..}}] → ["id", …] in the sorted order
const char* ph_engine_sort(void* h, const char* sort_name, const char* items_json, int size) {
    if (!h) { g_ph_err = "ph_engine_sort: NULL engine"; return nullptr; }
    Engine* e = (Engine*)h;
    json::Value root;
    std::string err;
    const std::string text = items_json ? items_json : "";
    if (!json::Parser(text).Parse(&root, &err)) { g_ph_err = err; return nullptr; }
    auto s = e->sorts.Get(sort_name ? sort_name : "");
    if (!s) { g_ph_err = std::string("Sort:not find, name:") + (sort_name ? sort_name : ""); return nullptr; }
    sort::SortData sd;
    context::RecommendContext ctx;
    ctx.Size = size;
    sd.Context = &ctx;
    for (const auto& it : root.arr) {
        auto item = std::make_shared<module::Item>(it.s("id"));
        item->Score = it.d("score");
        item->Properties = it.at("properties").obj;
        for (const auto& kv : it.at("algo_scores").obj) item->AddAlgoScore(kv.first, kv.second.num);
        sd.Data.push_back(item);
    }
    if (!s->Sort(&sd, &err)) { g_ph_err = err; return nullptr; }
    std::string& o = g_ph_out;
    o = "[";
    for (size_t i = 0; i < sd.Data.size(); ++i) { if (i) o += ","; json::Escape(sd.Data[i]->Id, &o); }
    o += "]";
    return o.c_str();
}

const char* ph_recommend(void* h, const char* uid, int size, const char* scene) {
    if (!h) return nullptr;
    std::vector<module::ItemPtr> items;
    std::string err;
    if (!((Engine*)h)->Recommend(uid ? uid : "", size, scene ? scene : "", &items, &err)) {
        g_ph_err = err;
        return nullptr;
    }
    return items_to_json(items);
}

// the same request with an AB experiment attached: experiment_params_json = the layer params object
const char* ph_recommend_ab(void* h, const char* uid, int size, const char* scene, const char* experiment_params_json) {
    if (!h) return nullptr;
    json::Value params;
    std::string err;
    const std::string text = experiment_params_json ? experiment_params_json : "{}";
    if (!json::Parser(text).Parse(&params, &err) || params.type != json::Value::Object) {
        g_ph_err = "experiment params: " + (err.empty() ? std::string("not a JSON object") : err);
        return nullptr;
    }
    std::vector<module::ItemPtr> items;
    if (!((Engine*)h)->Recommend(uid ? uid : "", size, scene ? scene : "", params, &items, &err)) {
        g_ph_err = err;
        return nullptr;
    }
    return items_to_json(items);
}

// n_threads threads, each serving uids[t], uids[t + n_threads], … one request at a time (pairec's goroutines): the
// engine's plug-ins are called concurrently, and with "Coalesce" configured the library batches them.  → JSON array of the
// pages in uid order
const char* ph_recommend_concurrent(void* h, const char* uids_json, int size, const char* scene, int n_threads) {
    if (!h) return nullptr;
    Engine* e = (Engine*)h;
    json::Value uids;
    std::string err;
    const std::string text = uids_json ? uids_json : "[]";
    if (!json::Parser(text).Parse(&uids, &err) || uids.type != json::Value::Array) { g_ph_err = "uids: " + err; return nullptr; }
    const size_t n = uids.arr.size();
    std::vector<std::vector<module::ItemPtr>> pages(n);
    std::vector<std::string> errs(n);
    std::vector<std::thread> th;
    const std::string sc = scene ? scene : "";
    for (int t = 0; t < n_threads; ++t)
        th.emplace_back([&, t]() {
            for (size_t i = (size_t)t; i < n; i += (size_t)n_threads)
                if (!e->Recommend(uids.arr[i].str, size, sc, &pages[i], &errs[i]) && errs[i].empty()) errs[i] = "failed";
        });
    for (auto& x : th) x.join();
    for (size_t i = 0; i < n; ++i)
        if (!errs[i].empty()) { g_ph_err = errs[i]; return nullptr; }
    std::string o = "[";
    for (size_t i = 0; i < n; ++i) {
        if (i) o += ",";
        o += items_to_json(pages[i]);
    }
    o += "]";
    g_ph_out = o;
    return g_ph_out.c_str();
}

// recall.Load's outcome for one RecallConfs entry: "built" | "panic: <text>" | "unavailable: <text>"
const char* ph_check_recall_conf(const char* conf_json) {
    recconf::RecommendConfig c;
    std::string err;
    const std::string text = std::string("{\"RecallConfs\":[") + (conf_json ? conf_json : "{}") + "]}";
    if (!recconf::RecommendConfig::Parse(text, &c, &err) || c.RecallConfs.empty()) { g_ph_err = err; return nullptr; }
    const recall::LoadOutcome o = recall::CheckRecallConf(c.RecallConfs[0]);
    g_ph_out = o.kind == recall::LoadOutcome::kBuilt ? "built" : (o.kind == recall::LoadOutcome::kPanic ? "panic: " : "unavailable: ") + o.message;
    return g_ph_out.c_str();
}

// response decoders driven by JSON (tests): {"func": name, ...inputs} → [{"score":..,"score_map":{..},"module_type":b,"classify":{..}}]
const char* ph_decode_response(const char* spec_json) {
    json::Value sp;
    std::string err;
    const std::string text = spec_json ? spec_json : "";
    if (!json::Parser(text).Parse(&sp, &err)) { g_ph_err = err; return nullptr; }
    const std::string fn = sp.s("func");
    std::vector<std::string> ids, outs;
    for (const auto& v : sp.at("item_ids").arr) ids.push_back(v.str);
    for (const auto& v : sp.at("outputs").arr) outs.push_back(v.str);
    std::map<std::string, std::vector<double>> results;
    for (const auto& kv : sp.at("results").obj) for (const auto& x : kv.second.arr) results[kv.first].push_back(x.num);
    std::vector<algorithm::AlgoResponse> ret;
    if (fn == "easyrecResponseFunc") ret = algorithm::decode::EasyrecResponse(ids, results);
    else if (fn == "easyrecMutValResponseFunc") {
        if (!algorithm::decode::EasyrecMutValResponse(ids, outs, results, &ret, &err)) { g_ph_err = err; return nullptr; }
    } else if (fn == "easyrecMutClassificationResponseFunc") {
        std::map<std::string, std::pair<std::vector<float>, std::vector<long long>>> tf;
        for (const auto& kv : sp.at("tf_outputs").obj) {
            auto& dst = tf[kv.first];
            for (const auto& x : kv.second.at("float_val").arr) dst.first.push_back((float)x.num);
            for (const auto& x : kv.second.at("shape").arr) dst.second.push_back((long long)x.num);
        }
        if (!algorithm::decode::EasyrecMutClassificationResponse(ids, tf, &ret, &err)) { g_ph_err = err; return nullptr; }
    } else if (fn == "alinkFMResponseFunc") {
        for (const auto& v : sp.at("predictions").arr)
            ret.emplace_back(algorithm::decode::AlinkFMScore(v.d("prediction_result"), v.d("prediction_score")));
    } else if (fn == "tfservingResponseFunc") {
        std::vector<std::vector<double>> o2;
        for (const auto& row : sp.at("tf_rows").arr) { o2.emplace_back(); for (const auto& x : row.arr) o2.back().push_back(x.num); }
        ret = algorithm::decode::TFServingResponse(o2);
    } else if (fn == "tfResponseFunc" || fn == "tfMutValResponseFunc" || fn == "torchrecMutValResponseFunc" ||
               fn == "torchrecMutValResponseFuncDebug" || fn == "torchrecMutClassificationResponseFunc" ||
               fn == "torchrecMutClassificationResponseFuncDebug" || fn == "torchrecEmbeddingItemsResponseFunc") {
        // "outputs_list": [{"name":…, "dtype": "float"|"double", "values": […], "shape": […]}] in the message's order
        std::vector<std::pair<std::string, algorithm::decode::OutputArray>> outsl;
        for (const auto& ov : sp.at("outputs_list").arr) {
            algorithm::decode::OutputArray oa;
            oa.is_double = ov.s("dtype") == "double";
            for (const auto& x : ov.at("values").arr) { if (oa.is_double) oa.double_val.push_back(x.num); else oa.float_val.push_back((float)x.num); }
            for (const auto& x : ov.at("shape").arr) oa.shape.push_back((long long)x.num);
            outsl.emplace_back(ov.s("name"), std::move(oa));
        }
        bool ok = true;
        if (fn == "tfResponseFunc") ret = algorithm::decode::TfResponse(outsl);
        else if (fn == "tfMutValResponseFunc") ret = algorithm::decode::TfMutValResponse(outsl);
        else if (fn.rfind("torchrecMutVal", 0) == 0) ok = algorithm::decode::TorchrecMutValResponse(ids.size(), outsl, &ret, &err);
        else if (fn.rfind("torchrecMutClassification", 0) == 0) ok = algorithm::decode::TorchrecMutClassificationResponse(ids.size(), outsl, &ret, &err);
        else {
            const algorithm::decode::OutputArray* sc = nullptr;
            for (const auto& kv : outsl) if (kv.first == "match_item_scores") sc = &kv.second;
            std::vector<algorithm::EmbeddingInfo> items;
            ok = algorithm::decode::TorchrecEmbeddingItemsResponse(ids, sc, &items, &err);
            std::string o = "[";
            for (size_t i = 0; ok && i < items.size(); ++i) {
                if (i) o += ",";
                o += "{\"item_id\":";
                json::Escape(items[i].ItemId, &o);
                char buf[64];
                snprintf(buf, sizeof buf, ",\"score\":%.17g}", items[i].Score);
                o += buf;
            }
            o += "]";
            if (!ok) { g_ph_err = err; return nullptr; }
            g_ph_out = o;
            return g_ph_out.c_str();
        }
        if (!ok) { g_ph_err = err; return nullptr; }
    } else if (fn == "easyrecResponseFuncDebug") {
        ret = algorithm::decode::EasyrecResponse(ids, results);                    // (:237-268: the same scores + the debug strings)
    } else if (fn == "easyrecMutValResponseFuncDebug") {
        if (!algorithm::decode::EasyrecMutValResponse(ids, outs, results, &ret, &err)) { g_ph_err = err; return nullptr; }
    } else if (fn == "pssmartResponseFunc") {
        for (const auto& v : sp.at("predictions").arr)
            ret.emplace_back(algorithm::decode::PssmartScore(v.d("score"), v.s("lable"), v.s("label")));
    } else if (fn == "widenF32") {
        std::vector<float> f;
        for (const auto& x : sp.at("float_val").arr) f.push_back((float)x.num);
        ret = algorithm::decode::WidenF32(f.data(), f.size());
    } else { g_ph_err = "unknown decoder " + fn; return nullptr; }
    std::string o = "[";
    for (size_t i = 0; i < ret.size(); ++i) {
        if (i) o += ",";
        o += "{\"score\":" + json::NumToString(ret[i].GetScore()) + ",\"module_type\":" + (ret[i].GetModuleType() ? "true" : "false") + ",\"score_map\":{";
        bool first = true;
        for (const auto& kv : ret[i].GetScoreMap()) {
            if (!first) o += ",";
            first = false;
            json::Escape(kv.first, &o);
            o += ":" + json::NumToString(kv.second);
        }
        o += "},\"classify\":{";
        first = true;
        for (const auto& kv : ret[i].GetClassifyMap()) {
            if (!first) o += ",";
            first = false;
            json::Escape(kv.first, &o);
            o += ":[";
            for (size_t c = 0; c < kv.second.size(); ++c) o += (c ? "," : "") + json::NumToString(kv.second[c]);
            o += "]";
        }
        o += "}}";
    }
    o += "]";
    g_ph_out = o;
    return g_ph_out.c_str();
}

// fmt %v of a float64
const char* ph_go_fmt_float(double x) {
    g_ph_out = GoFmtFloat(x);
    return g_ph_out.c_str();
}

// recall result cache line: items JSON [{"id":..,"score":..}] → "id:name:score,…"
const char* ph_format_recall_cache(const char* items_json, const char* recall_name) {
    json::Value root;
    std::string err;
    const std::string text = items_json ? items_json : "";
    if (!json::Parser(text).Parse(&root, &err) || root.type != json::Value::Array) { g_ph_err = "items: " + err; return nullptr; }
    std::vector<module::ItemPtr> items;
    for (const auto& v : root.arr) {
        auto it = std::make_shared<module::Item>(v.s("id"));
        it->Score = v.d("score");
        items.push_back(it);
    }
    g_ph_out = recall::FormatCacheString(items, recall_name ? recall_name : "");
    return g_ph_out.c_str();
}

// … and back: "id:name:score,…" → items JSON; NULL (+ ph_last_error) where the reference would panic
const char* ph_parse_recall_cache(const char* line, const char* recall_name, const char* item_type) {
    std::vector<module::ItemPtr> items;
    std::string err;
    if (!recall::ParseCacheString(line ? line : "", recall_name ? recall_name : "", item_type ? item_type : "", &items, &err)) {
        g_ph_err = err;
        return nullptr;
    }
    return items_to_json(items);
}

// EasyrecAlgoDataGenerator driver: {"context_features":[names],"item_features":[names]|null,
// "user":{..},"items":[{"id":..,"features":{..}}], "batches": [n1, n2, ...]} → [request JSON per batch]
const char* ph_easyrec_generator(const char* spec_json) {
    json::Value root;
    std::string err;
    const std::string text = spec_json ? spec_json : "";
    if (!json::Parser(text).Parse(&root, &err)) { g_ph_err = err; return nullptr; }
    std::vector<std::string> ctxf, inf;
    for (const auto& v : root.at("context_features").arr) ctxf.push_back(v.str);
    rank::EasyrecAlgoDataGenerator g(ctxf);
    if (root.at("item_features").type == json::Value::Array) {
        for (const auto& v : root.at("item_features").arr) inf.push_back(v.str);
        g.SetItemFeatures(inf);
    }
    std::map<std::string, json::Value> user(root.at("user").obj.begin(), root.at("user").obj.end());
    std::string& o = g_ph_out;
    o = "[";
    size_t pos = 0;
    bool first = true;
    for (const auto& b : root.at("batches").arr) {
        for (long long k = 0; k < (long long)b.num && pos < root.at("items").arr.size(); ++k, ++pos) {
            const json::Value& it = root.at("items").arr[pos];
            std::map<std::string, json::Value> feats(it.at("features").obj.begin(), it.at("features").obj.end());
            g.AddFeatures(std::make_shared<module::Item>(it.s("id")), feats, user);
        }
        if (!first) o += ",";
        first = false;
        o += g.GeneratorAlgoData();
    }
    o += "]";
    return o.c_str();
}

// ---- service/feature drivers -------------------------------------------------------------------------------------
static void dyn_json(const json::Value& v, std::string* o) {        // floats keep a '.', so a reader can tell float64 10 from int 10
    switch (v.type) {
        case json::Value::String: json::Escape(v.str, o); break;
        case json::Value::Bool: *o += v.b ? "true" : "false"; break;
        case json::Value::Number:
            if (v.is_u64) *o += std::to_string((unsigned long long)v.i);
            else if (v.is_int) *o += std::to_string(v.i);
            else if (v.num != v.num || std::isinf(v.num)) *o += "null";
            else {
                std::string t = json::NumToString(v.num);
                if (t.find_first_of(".eEn") == std::string::npos) t += ".0";
                *o += t;
            }
            break;
        case json::Value::Array:
            *o += "[";
            for (size_t i = 0; i < v.arr.size(); ++i) { if (i) *o += ","; dyn_json(v.arr[i], o); }
            *o += "]";
            break;
        case json::Value::Object: {
            *o += "{";
            bool first = true;
            for (const auto& kv : v.obj) { if (!first) *o += ","; first = false; json::Escape(kv.first, o); *o += ":"; dyn_json(kv.second, o); }
            *o += "}";
            break;
        }
        default: *o += "null";
    }
}
static const char* go_type_of(const json::Value& v) {
    switch (v.type) {
        case json::Value::String: return "string";
        case json::Value::Bool: return "bool";
        case json::Value::Number: return v.is_u64 ? "uint64" : (v.is_int ? "int" : "float64");
        case json::Value::Array: return "slice";
        case json::Value::Object: return "map";
        default: return "nil";
    }
}
// NewNormalizer(name, expression).Apply(value): {"name":..,"expression":..,"value":..,"clock_ms":n?} →
// {"kind": Go type of the normalizer | null, "result": .., "type": Go type of the result}; an expression outside the subset → NULL + error
const char* ph_normalizer_apply(const char* spec_json) {
    json::Value root;
    std::string err;
    const std::string text = spec_json ? spec_json : "";
    if (!json::Parser(text).Parse(&root, &err)) { g_ph_err = err; return nullptr; }
    feature::SetClockForTest(root.n("clock_ms", 0));
    auto nz = feature::NewNormalizer(root.s("name"), root.s("expression"), &err);
    if (!nz && !err.empty()) { g_ph_err = err; return nullptr; }
    std::string& o = g_ph_out;
    if (!nz) { o = "{\"kind\":null}"; return o.c_str(); }
    const json::Value r = nz->Apply(root.at("value"));
    o = std::string("{\"kind\":\"") + nz->Kind() + "\",\"type\":\"" + go_type_of(r) + "\",\"result\":";
    dyn_json(r, &o);
    o += "}";
    return o.c_str();
}
// Feature.LoadFeatures: {"features":[FeatureConfig…], "user":{"id":..,"properties":{..}}|null,
// "items":[{"id":..,"retrieve_id":..,"properties":{..}}], "context_features":{..}?, "clock_ms":n?} → {"user":{..}|null,"items":[{..}]}
const char* ph_feature_load(const char* spec_json) {
    json::Value root;
    std::string err;
    const std::string text = spec_json ? spec_json : "";
    if (!json::Parser(text).Parse(&root, &err)) { g_ph_err = err; return nullptr; }
    feature::SetClockForTest(root.n("clock_ms", 0));
    std::vector<feature::FeatureConfig> confs;
    for (const auto& f : root.at("features").arr) {
        feature::FeatureConfig c;
        c.FeatureType = f.s("FeatureType"); c.FeatureName = f.s("FeatureName"); c.FeatureSource = f.s("FeatureSource");
        c.FeatureValue = f.s("FeatureValue"); c.FeatureStore = f.s("FeatureStore"); c.Normalizer = f.s("Normalizer");
        c.Expression = f.s("Expression");
        c.RemoveFeatureSource = f.at("RemoveFeatureSource").type == json::Value::Bool && f.at("RemoveFeatureSource").b;
        confs.push_back(std::move(c));
    }
    feature::Feature feat;
    if (!feat.LoadWithConfig(confs, &err)) { g_ph_err = err; return nullptr; }
    std::unique_ptr<module::User> user;
    if (root.at("user").type == json::Value::Object) {
        user.reset(new module::User(root.at("user").s("id")));
        user->Properties = root.at("user").at("properties").obj;
    }
    std::vector<module::ItemPtr> items;
    for (const auto& it : root.at("items").arr) {
        auto item = std::make_shared<module::Item>(it.s("id"));
        item->RetrieveId = it.s("retrieve_id");
        item->Properties = it.at("properties").obj;
        items.push_back(item);
    }
    context::RecommendContext ctx;
    if (root.at("context_features").type == json::Value::Object) ctx.Param["features"] = root.at("context_features");
    feat.LoadFeatures(user.get(), items, &ctx);
    std::string& o = g_ph_out;
    o = "{\"user\":";
    if (user) { json::Value u; u.type = json::Value::Object; u.obj = user->Properties; dyn_json(u, &o); } else o += "null";
    o += ",\"items\":[";
    for (size_t i = 0; i < items.size(); ++i) {
        if (i) o += ",";
        json::Value p; p.type = json::Value::Object; p.obj = items[i]->Properties;
        dyn_json(p, &o);
    }
    o += "]}";
    return o.c_str();
}

// cache adapters + clone hooks, self-checked in C++ (bit i set = check i passed)
int ph_cache_clone_semantics(void) {
    int ok = 0;
    std::string err;
    auto lc = cache::NewCache("localCache", "", &err);
    auto bc = cache::NewCache("localBytes", "", &err);
    if (lc && bc && !cache::NewCache("memcached", "", &err) && err == "Cache:not found instance, name:memcached") ok |= 1;
    if (lc && bc) {
        lc->Put("k", "a:r:1", 1800);
        bc->Put("k", "a:r:1", 1800);
        // localCache hands back a string (VectorRecall's []uint8 assertion misses), the byte store hits
        if (lc->Get("k").kind == cache::Value::kString && bc->Get("k").kind == cache::Value::kBytes) ok |= 2;
        if (bc->Get("absent").kind == cache::Value::kNone) ok |= 4;
        bc->Put("gone", "x", 0 + 1);
    }
    // ICloneSort: the hook swaps in the clone only when the experiment carries a "sort.<name>" object
    struct Rev : sort::ISort, sort::ICloneSort {
        bool reversed;
        explicit Rev(bool r) : reversed(r) {}
        bool Sort(sort::SortData* d, std::string*) override {
            if (reversed) std::reverse(d->Data.begin(), d->Data.end());
            return true;
        }
        std::shared_ptr<sort::ISort> CloneWithConfig(const json::Value& p) override {
            return std::make_shared<Rev>(p.at("reverse").type == json::Value::Bool && p.at("reverse").b);
        }
        std::string GetSortName() const override { return "rev"; }
    };
    {
        auto base = std::make_shared<Rev>(false);
        json::Value params;
        const std::string ptext = "{\"sort.rev\":{\"reverse\":true}}";     // (Parser keeps a reference)
        json::Parser(ptext).Parse(&params, &err);
        std::shared_ptr<sort::ISort> s = base;
        if (auto* cs = dynamic_cast<sort::ICloneSort*>(s.get())) {
            const json::Value& sc = params.at("sort." + cs->GetSortName());
            if (sc.type == json::Value::Object)
                if (auto c = cs->CloneWithConfig(sc)) s = c;
        }
        sort::SortData d;
        d.Data = {std::make_shared<module::Item>("a"), std::make_shared<module::Item>("b")};
        s->Sort(&d, &err);
        if (d.Data[0]->Id == "b" && s != base) ok |= 8;
    }
    return ok;
}

// host-only helpers (no GPU): used by the CPU tests of the mirror
int ph_parse_vector_string(const char* s, float* out, int cap) {
    const auto v = recall::ParseVectorString(s ? s : "");
    for (int i = 0; i < (int)v.size() && i < cap; ++i) out[i] = v[i];
    return (int)v.size();
}

// in: JSON [{"id":..,"score":..,"retrieve_id":..,"algo_scores":{..}}] → UniqueFilter → same shape + recall_scores
const char* ph_unique_filter(const char* items_json) {
    json::Value root;
    std::string err;
    if (!json::Parser(items_json ? items_json : "").Parse(&root, &err)) { g_ph_err = err; return nullptr; }
    std::vector<module::ItemPtr> items;
    for (const auto& v : root.arr) {
        auto it = std::make_shared<module::Item>(v.s("id"));
        it->Score = v.d("score");
        it->RetrieveId = v.s("retrieve_id");
        for (const auto& kv : v.at("algo_scores").obj) it->AddAlgoScore(kv.first, kv.second.num);
        items.push_back(it);
    }
    const auto out = filter::UniqueFilter(items);
    std::string& o = g_ph_out;
    o = "[";
    for (size_t i = 0; i < out.size(); ++i) {
        if (i) o += ",";
        o += "{\"id\":";
        json::Escape(out[i]->Id, &o);
        o += ",\"score\":" + json::NumToString(out[i]->Score) + ",\"algo_scores\":{";
        bool first = true;
        for (const auto& kv : out[i]->algoScores) {
            if (!first) o += ",";
            first = false;
            json::Escape(kv.first, &o);
            o += ":" + json::NumToString(kv.second);
        }
        o += "},\"recall_scores\":{";
        first = true;
        for (const auto& kv : out[i]->RecallScores) {
            if (!first) o += ",";
            first = false;
            json::Escape(kv.first, &o);
            o += ":" + json::NumToString(kv.second);
        }
        o += "}}";
    }
    o += "]";
    return o.c_str();
}

// registry semantics: returns a bitmask of checks that hold (all 5 bits set = 31)
int ph_registry_semantics(void) {
    struct DummySort : sort::ISort { int id; explicit DummySort(int i) : id(i) {} bool Sort(sort::SortData*, std::string*) override { return true; } };
    struct DummyAlgo : algorithm::IAlgorithm {
        double v; explicit DummyAlgo(double x) : v(x) {}
        bool Init(const recconf::AlgoConfig&, std::string*) override { return true; }
        bool Run(const algorithm::AlgoData&, algorithm::AlgoResult* out, std::string*) override { out->responses.assign(1, algorithm::AlgoResponse(v)); return true; }
    };
    int ok = 0;
    sort::Registry sr;
    std::string err;
    auto s1 = std::make_shared<DummySort>(1), s2 = std::make_shared<DummySort>(2);
    sr.RegisterSort("x", s1, &err);
    sr.RegisterSort("x", s2, &err);
    if (sr.Get("x") == s1) ok |= 1;                                   // first registration wins
    if (!sr.RegisterSort("nil", nullptr, &err)) ok |= 2;              // nil rejected (reference panics)
    algorithm::AlgorithmFactory f;
    f.RegisterAlgorithm("a", std::make_shared<DummyAlgo>(1.0));
    f.RegisterAlgorithm("a", std::make_shared<DummyAlgo>(2.0));
    algorithm::AlgoResult res;
    if (f.Run("a", algorithm::AlgoData{}, &res, &err) && res.responses[0].GetScore() == 2.0) ok |= 4;   // overwrite
    if (!f.Run("missing", algorithm::AlgoData{}, &res, &err) && err.find("not find algorithm") != std::string::npos) ok |= 8;
    recall::Registry rr;
    if (!rr.GetRecall("nope", &err)) ok |= 16;
    return ok;
}

// parse a recconf JSON and echo the subset the engine honours (for config tests)
const char* ph_parse_recconf(const char* text) {
    recconf::RecommendConfig c;
    std::string err;
    if (!recconf::RecommendConfig::Parse(text ? text : "", &c, &err)) { g_ph_err = err; return nullptr; }
    std::string& o = g_ph_out;
    o = "{\"recalls\":" + std::to_string(c.RecallConfs.size()) + ",\"gpu_recalls\":" + std::to_string(c.GpuRecalls.size()) +
        ",\"gpu_sorts\":" + std::to_string(c.GpuSorts.size()) + ",\"algos\":" + std::to_string(c.AlgoConfs.size()) +
        ",\"rank_scenes\":" + std::to_string(c.RankConf.size()) + ",\"dpp\":" + std::to_string(c.DPPConf.size());
    auto count_features = [](const std::map<std::string, recconf::SceneFeatureConfig>& m) {
        size_t n = 0;
        for (const auto& sc : m) for (const auto& lc : sc.second.FeatureLoadConfs) n += lc.Features.size();
        return n;
    };
    if (!c.FeatureConfs.empty() || !c.UserFeatureConfs.empty())
        o += ",\"feature_transforms\":" + std::to_string(count_features(c.FeatureConfs)) + ",\"user_feature_transforms\":" +
             std::to_string(count_features(c.UserFeatureConfs));
    auto echo = [&](const char* key, const recconf::RecallConfig& r) {
        o += std::string(",\"") + key + "\":{\"name\":";
        json::Escape(r.Name, &o);
        o += ",\"count\":" + std::to_string(r.RecallCount) + ",\"algo\":";
        json::Escape(r.RecallAlgo, &o);
        o += "}";
    };
    if (!c.RecallConfs.empty()) echo("recall0", c.RecallConfs[0]);
    if (!c.GpuRecalls.empty()) echo("gpu_recall0", c.GpuRecalls[0]);
    for (const auto& kv : c.RankConf) {
        o += ",\"rank_" + kv.first + "\":{\"batch\":" + std::to_string(kv.second.BatchCount) + ",\"score\":";
        json::Escape(kv.second.RankScore, &o);
        o += ",\"algos\":" + std::to_string(kv.second.RankAlgoList.size()) + ",\"score_rewrite\":{";
        bool first = true;
        for (const auto& rw : kv.second.ScoreRewrite) {
            if (!first) o += ",";
            first = false;
            json::Escape(rw.first, &o);
            o += ":";
            json::Escape(rw.second, &o);
        }
        o += "}}";
    }
    o += "}";
    return o.c_str();
}

}  // extern "C"
