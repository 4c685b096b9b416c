// pairec_host.hpp — C++ mirror of pairec's plugin surfaces for the rank + recall hot path.
//
// The reference is Go and this image has no Go toolchain, so the host side above the C ABI
// (include/pairec_gpu.h) is written in C++ with the reference's names, argument meaning and error
// behaviour, so that the tests read like the reference's own:
//   module::Item / User            module/item.go:15-27, module/user.go:16-25
//   context::RecommendContext      context/recommend_context.go:18-34
//   algorithm::IAlgorithm + factory algorithm/algorithm.go:28-31,107-120,164-168
//   recall::Recall + registry      service/recall/recall.go:18-20,32-45
//   sort::ISort + registry         sort/sort.go:33-35,143-150
//   filter::UniqueFilter           filter/unique_filter.go:26-49
//   rank::RankService              service/rank/rank_service.go:102-372
//   recconf subset                 recconf/recconf.go:48-93,267-276,330-367,736-745,960-979
// The GPU-backed plugins (GpuFaissAlgorithm, GpuDnnAlgorithm, GpuVectorRecall, Gpu*Sort) are what
// the cgo shim of INTEGRATION.md registers under the same registries in a real pairec process.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pairec_gpu.h"
#include "json.hpp"

namespace pairec {

// ---- module -------------------------------------------------------------------------------------
namespace module {
using ItemId = std::string;

struct Item {
    ItemId Id;
    double Score = 0.0;
    std::string RetrieveId;
    std::string ItemType;
    std::vector<double> Embedding;
    std::map<std::string, json::Value> Properties;
    std::map<std::string, double> algoScores;
    std::map<std::string, double> RecallScores;
    bool hasRecallScores = false;

    explicit Item(ItemId id = "") : Id(std::move(id)) {}
    void AddAlgoScore(const std::string& name, double score) { algoScores[name] = score; }   // item.go:168-176
    void AddAlgoScores(const std::map<std::string, double>& scores) { for (const auto& kv : scores) algoScores[kv.first] = kv.second; }   // item.go:177-188
    void AddProperty(const std::string& k, json::Value v) { Properties[k] = std::move(v); }
    // Item.FloatExprData (item.go:189-212): "current_score" has the recall_score side effect
    bool FloatExprData(const std::string& name, double* out);
};
using ItemPtr = std::shared_ptr<Item>;

struct User {
    std::string Id;
    std::map<std::string, json::Value> Properties;
    explicit User(std::string id = "") : Id(std::move(id)) {}
};

// module.VectorDao (vector_dao.go:13-15): returns "1:v1 2:v2 …"; empty → VectoryEmptyError
struct VectorDao {
    virtual ~VectorDao() = default;
    virtual bool VectorString(const std::string& id, std::string* out, std::string* err) = 0;
};
struct InMemoryVectorDao : VectorDao {
    std::map<std::string, std::string> vectors;
    bool VectorString(const std::string& id, std::string* out, std::string* err) override;
};
}  // namespace module

// utils.ToFloat (utils/type.go:43-69)
double ToFloat(const json::Value& v, double def);

namespace context {
struct RecommendContext {
    int Size = 10;
    bool Debug = false;
    std::string RecommendId;
    std::map<std::string, json::Value> Param;               // GetParameter("scene") etc.
    std::map<std::string, double> ExperimentParams;         // AB overrides (float view)
    // ExperimentResult.GetExperimentParams() (model.LayerParams): the raw JSON object — structured
    // values such as "recall.<name>" / "sort.<name>" objects or "ssd_filter_retrieve_ids" lists
    json::Value ExperimentParamsJson;
    bool HasExperiment() const { return ExperimentParamsJson.type == json::Value::Object; }
    double GetFloat(const std::string& k, double def) const;   // LayerParams.GetFloat
    long long GetInt(const std::string& k, long long def) const;
    std::string GetParameter(const std::string& k) const;
};
}  // namespace context

// ---- recconf (subset) ---------------------------------------------------------------------------
namespace recconf {
struct AlgoConfig { std::string Name, Type; json::Value raw; };
struct RecallConfig {
    std::string Name, RecallType, RecallAlgo, ItemType, CacheAdapter, CacheConfig, CachePrefix;
    int RecallCount = 0, CacheTime = 0;
    // what the reference's constructors look at before they touch a datasource (recconf.go:330-367)
    std::string DaoAdapterType, VectorDaoAdapterType, HologresName, VectorAlgoType;
    std::string Kind;                       // pairec_gpu.Recalls only: "vector" (default), "i2i", "hologres", "hologres_v2", "page"
    // HologresVectorConf.WhereClause / TimeInterval (recconf.go:492-497), in the form the device serves: `column OP constant`
    std::string WhereClause, WhereColumn;
    int WhereOp = -1;                       // pg_where_op
    long long WhereValue = 0;
    int TimeInterval = 0;
    std::string RankScore, RankVar;         // Kind "page": the RankScore expression and the name the model's score has in it
    std::map<std::string, std::string> ScoreRewrite;     // Kind "page": RankConfig.ScoreRewrite of the scene it finishes (evaluated on the device)
};
struct RankConfig {
    std::vector<std::string> RankAlgoList;
    std::string RankScore, Processor, ASTType;
    int BatchCount = 0;
    std::map<std::string, std::string> ScoreRewrite;     // recconf.go:743: algo-score name → expression (rank_service.go:296-306,343-353)
};
struct DPPSortConfig {
    std::string Name;
    double Alpha = 1.0;
    int WindowSize = 0, CandidateCount = 0, AbortRunCount = 0;
    double MinScorePercent = 0.0;
    bool NormalizeEmb = true, EnsurePositiveSim = true;
    std::vector<std::string> FilterRetrieveIds, EmbeddingHookNames;
};
struct SSDSortConfig {                     // recconf.go:980-1000 (the fields the device path consumes)
    std::string Name;
    double Gamma = 0.25;                   // NewSSDSort: 0.25 unless Gamma > 0 (ssd_sort.go:68,81-83)
    bool UseSSDStar = false;
    bool NormalizeEmb = true, EnsurePositiveSim = true;
    int WindowSize = 5, AbortRunCount = 0, CandidateCount = 0;
    double MinScorePercent = 0.0;
    std::vector<std::string> FilterRetrieveIds;
};
struct SortConfig {                        // recconf.go:820-838: Name, SortType, nested DPPConf / SSDConf
    std::string Name, SortType;
    std::string SortByField;               // AlgoScoreSort (sort/algo_score_sort.go:17-27), CustomFieldSort (custom_field_sort.go:21-38)
    std::string SortOrder;                 // CustomFieldSort: "asc" | "desc" (anything else is "desc")
    double SwitchThreshold = 0.0;
    std::string HologresName;              // DPPConf / SSDConf .DaoConf.HologresName: what NewDPPSort / NewSSDSort open first
    DPPSortConfig DPPConf;
    SSDSortConfig SSDConf;
};
struct FeatureConfig {                      // recconf.go:256-265
    std::string FeatureType, FeatureName, FeatureSource, FeatureValue, FeatureStore, Normalizer, Expression;
    bool RemoveFeatureSource = false;
};
struct FeatureLoadConfig { std::vector<FeatureConfig> Features; std::string DaoAdapterType; };   // recconf.go:173-176 (the DAO fetch is storage: out of scope)
struct SceneFeatureConfig { std::vector<FeatureLoadConfig> FeatureLoadConfs; bool AsynLoadFeature = false; };   // recconf.go:169-172
struct RecommendConfig {
    std::map<std::string, SceneFeatureConfig> UserFeatureConfs, FeatureConfs;   // by scene (recconf.go:52-53)
    std::vector<AlgoConfig> AlgoConfs;
    std::vector<RecallConfig> RecallConfs;
    std::map<std::string, RankConfig> RankConf;                               // by scene
    std::map<std::string, std::vector<std::string>> SortNames;                // by scene
    std::map<std::string, std::map<std::string, std::vector<std::string>>> SceneRecallNames;  // scene → category → RecallNames
    std::vector<DPPSortConfig> DPPConf;
    std::vector<SortConfig> SortConfs;                                       // recconf.go:86
    std::vector<RecallConfig> GpuRecalls;                                    // UserDefineConfs.pairec_gpu.Recalls
    std::vector<SortConfig> GpuSorts;                                        // UserDefineConfs.pairec_gpu.Sorts
    json::Value UserDefineConfs;                                              // recconf.go:92
    static bool Parse(const std::string& text, RecommendConfig* out, std::string* err);
};
}  // namespace recconf

// ---- algorithm ----------------------------------------------------------------------------------
namespace algorithm {
// what IAlgorithm.Run receives / returns on the hot path (algoData is interface{} in Go)
struct VectorRequest { uint32_t K = 0; std::vector<float> Vector; };                 // pai_web.VectorRequest
struct VectorReply { std::vector<uint64_t> Retval; std::vector<float> Scores; std::vector<std::string> Labels; };
struct RankRequest {                        // the GPU flavour of []map[string]interface{} / PBRequest:
    std::vector<float> UserVector;          // user features already reduced to the model's user vector
    std::vector<std::string> ItemIds;       // request order = response order (rank_service.go:312-335)
    std::vector<int32_t> UserFieldIds;      // EasyRec flavour: the user's categorical features, dictionary-encoded
};
struct EmbeddingRequest {                   // the *easyrec.PBRequest an OnlineVectorRecall sends (online_vector_recall.go:97-109)
    std::vector<float> UserVector;          // user_features
    int FaissNeighNum = 0;
};
struct AlgoResponse {                       // response.AlgoResponse (algorithm/response/resonse.go:3-7)
    double score = 0.0;
    std::map<std::string, double> scoreArr; // multi-output models: output name → score (EasyrecResponse.scoreArr)
    bool multiValModule = false;
    // response.AlgoMultiClassifyResponse (resonse.go:9-11): output name → class probabilities
    std::map<std::string, std::vector<double>> mulClassifyArr;
    AlgoResponse() = default;
    explicit AlgoResponse(double s) : score(s) {}
    double GetScore() const { return score; }
    const std::map<std::string, double>& GetScoreMap() const { return scoreArr; }
    bool GetModuleType() const { return multiValModule; }
    bool IsMultiClassify() const { return !mulClassifyArr.empty(); }
    const std::map<std::string, std::vector<double>>& GetClassifyMap() const { return mulClassifyArr; }
};
// eas.EmbeddingInfo (algorithm/eas/easyrec_response.go:676-680): what a vector model answers an
// OnlineVectorRecall with (TorchrecEmbeddingItemsResponse)
struct EmbeddingInfo { std::string ItemId; double Score = 0.0; };

// ---- response decoders (the ResponseFunc family): remote model output → []AlgoResponse in request order ----------
namespace decode {
// easyrecResponseFunc (algorithm/eas/easyrec_response.go:220-236): Results[item_id].Scores[0]; a missing id scores 0
std::vector<AlgoResponse> EasyrecResponse(const std::vector<std::string>& item_ids,
                                          const std::map<std::string, std::vector<double>>& results);
// easyrecMutValResponseFunc (:35-70): one map {outputs[k]: Scores[k]} per item; a missing id maps every output to 0;
// a length mismatch is the reference's "outputs size is not equal scores" error
bool EasyrecMutValResponse(const std::vector<std::string>& item_ids, const std::vector<std::string>& outputs,
                           const std::map<std::string, std::vector<double>>& results,
                           std::vector<AlgoResponse>* out, std::string* err);
// easyrecMutClassificationResponseFunc (:145-218): float32 TfOutputs; shape [N] → one value per item, shape [N, C]
// → C values per item (easyrec_response_test.go:11-73)
bool EasyrecMutClassificationResponse(const std::vector<std::string>& item_ids,
                                      const std::map<std::string, std::pair<std::vector<float>, std::vector<long long>>>& tf_outputs,
                                      std::vector<AlgoResponse>* out, std::string* err);
// alinkFMResponse.GetScore (algorithm/eas/fm_response.go:28-34): prediction_result == 0 → 1 - prediction_score
double AlinkFMScore(double prediction_result, double prediction_score);
// tfservingResponseFunc (algorithm/tfserving/response.go:51-64): Outputs [][]float64 flattened row by row
std::vector<AlgoResponse> TFServingResponse(const std::vector<std::vector<double>>& outputs);
// tfResponseFunc / torchrecMutValResponseFunc's float32 → float64 widening (eas/tf_response.go:55-59)
std::vector<AlgoResponse> WidenF32(const float* scores, size_t n);
// one model output tensor as the PAI-EAS protobufs carry it: float32 or float64 values, row-major, with its shape
struct OutputArray {
    bool is_double = false;
    std::vector<float> float_val;
    std::vector<double> double_val;
    std::vector<long long> shape;
    double at(size_t i) const { return is_double ? double_val[i] : (double)float_val[i]; }
    size_t size() const { return is_double ? double_val.size() : float_val.size(); }
};
// tfResponseFunc (eas/tf_response.go:50-62): the FIRST output's float32 values, one score per item
std::vector<AlgoResponse> TfResponse(const std::vector<std::pair<std::string, OutputArray>>& outputs);
// tfMutValResponseFunc (eas/tf_response.go:29-48): item i's map {output name: FloatVal[i]} over every output
std::vector<AlgoResponse> TfMutValResponse(const std::vector<std::pair<std::string, OutputArray>>& outputs);
// torchrecMutValResponseFunc[Debug] (eas/easyrec_response.go:468-535): per item {output: FloatVal[i] | DoubleVal[i]}
bool TorchrecMutValResponse(size_t n_items, const std::vector<std::pair<std::string, OutputArray>>& outputs,
                            std::vector<AlgoResponse>* out, std::string* err);
// torchrecMutClassificationResponseFunc[Debug] (:537-626): shape [N] → one value, [N, C] → C values per item (float32)
bool TorchrecMutClassificationResponse(size_t n_items, const std::vector<std::pair<std::string, OutputArray>>& outputs,
                                       std::vector<AlgoResponse>* out, std::string* err);
// torchrecEmbeddingItemsResponseFunc (:700-734): item_ids[i] with match_item_scores[i] (float32 or float64)
bool TorchrecEmbeddingItemsResponse(const std::vector<std::string>& item_ids, const OutputArray* match_item_scores,
                                    std::vector<EmbeddingInfo>* out, std::string* err);
// pssmartResponse.GetScore (eas/pmml_response.go:10-32): label "0" (either spelling of the key) → 1 - score
double PssmartScore(double score, const std::string& lable, const std::string& label);
}  // namespace decode
struct AlgoData {
    enum Kind { kVector, kRank, kEmbedding } kind = kVector;
    VectorRequest vec;
    RankRequest rank;
    EmbeddingRequest emb;
};
struct AlgoResult {
    VectorReply reply;
    std::vector<AlgoResponse> responses;
    std::vector<EmbeddingInfo> embeddingItems;      // TorchrecEmbeddingItemsResponse.GetEmbeddingItems()
};

struct IAlgorithm {
    virtual ~IAlgorithm() = default;
    virtual bool Init(const recconf::AlgoConfig& conf, std::string* err) = 0;
    virtual bool Run(const AlgoData& data, AlgoResult* out, std::string* err) = 0;   // (interface{}, error)
};

// AlgorithmFactory (algorithm.go:33-120): RWMutex-guarded map, RegisterAlgorithm overwrites
class AlgorithmFactory {
public:
    void RegisterAlgorithm(const std::string& name, std::shared_ptr<IAlgorithm> a);
    bool Run(const std::string& name, const AlgoData& data, AlgoResult* out, std::string* err);
private:
    std::mutex mu_;
    std::map<std::string, std::shared_ptr<IAlgorithm>> algos_;
};
}  // namespace algorithm

// ---- recall -------------------------------------------------------------------------------------
namespace recall {
struct Recall {
    virtual ~Recall() = default;
    virtual std::vector<module::ItemPtr> GetCandidateItems(module::User* user, context::RecommendContext* ctx) = 0;
};
class Registry {                      // recalls map (recall.go:29-45): unguarded, overwrites
public:
    void RegisterRecall(const std::string& name, std::shared_ptr<Recall> r) { recalls_[name] = std::move(r); }
    std::shared_ptr<Recall> GetRecall(const std::string& name, std::string* err);
private:
    std::map<std::string, std::shared_ptr<Recall>> recalls_;
};
// vector_recall.go:70-82
std::vector<float> ParseVectorString(const std::string& s);
// `column OP integer` of a HologresVectorConf.WhereClause ("${time}" = now - time_interval, hologres_vector_recall.go:56-61);
// op: pg_where_op.  False for anything else (conjunctions, functions, strings, floats).
bool ParseWhereClause(const std::string& s, int time_interval, std::string* column, int* op, long long* value);
// recall.Load (service/recall/recall.go:47-107) over RecallConfs with the reference's outcomes: an unknown RecallType
// leaves the recall nil → panic("recall empty, name:…"); a type whose constructor opens a DAO / datasource that the
// entry does not configure panics inside that constructor ("not found VectorDao implement", …).  The mirror has no
// datasources, so a correctly configured datasource-backed entry is reported as unavailable rather than built.
// Returns false with *panic_msg = the reference's panic text (or the unavailability note).
struct LoadOutcome { enum Kind { kBuilt, kPanic, kUnavailable } kind = kBuilt; std::string message; };
LoadOutcome CheckRecallConf(const recconf::RecallConfig& conf);
// ICloneRecall (service/recall/recall.go:22-27): AB-experiment overrides ("recall.<name>" params object)
struct ICloneRecall {
    virtual ~ICloneRecall() = default;
    virtual std::shared_ptr<Recall> CloneWithConfig(const json::Value& params) = 0;
    virtual std::string GetRecallName() const = 0;
};
// the recall result cache line of VectorRecall (vector_recall.go:35-58,103-120): "id:name:score,…",
// scores printed with fmt's %v
std::string FormatCacheString(const std::vector<module::ItemPtr>& items, const std::string& recall_name);
bool ParseCacheString(const std::string& line, const std::string& recall_name, const std::string& item_type,
                      std::vector<module::ItemPtr>* out, std::string* err);
}  // namespace recall

// fmt.Sprintf("%v", float64): strconv 'g' with the shortest round-trip digits, exponent form for
// exp < -4 || exp >= 6
std::string GoFmtFloat(double x);

// ---- cache (persist/cache/cache.go:13-41) ---------------------------------------------------------
namespace cache {
// What Get returns matters to the caller: VectorRecall only accepts []uint8 (vector_recall.go:38), which
// the redis adapter returns and localCache (it hands back the Go string it was given) does not — with
// "localCache" the reference writes the line and never reads it back.  Mirrored: kString values miss.
struct Value { enum Kind { kNone, kString, kBytes } kind = kNone; std::string data; };
struct Cache {
    virtual ~Cache() = default;
    virtual void Put(const std::string& key, const std::string& val, int ttl_seconds) = 0;
    virtual Value Get(const std::string& key) = 0;
};
// adapters: "localCache" (persist/cache/localcache.go; string values) and "localBytes" (an in-process
// stand-in for the redis adapter's []byte values — this engine has no network); anything else is the
// reference's "Cache:not found instance" error
std::shared_ptr<Cache> NewCache(const std::string& adapter, const std::string& config, std::string* err);
}  // namespace cache

// ---- filter -------------------------------------------------------------------------------------
namespace filter {
std::vector<module::ItemPtr> UniqueFilter(const std::vector<module::ItemPtr>& items);   // unique_filter.go:26-49
}

// ---- sort ---------------------------------------------------------------------------------------
namespace sort {
struct SortData {
    std::vector<module::ItemPtr> Data;
    context::RecommendContext* Context = nullptr;
    module::User* User = nullptr;
};
struct ISort {
    virtual ~ISort() = default;
    virtual bool Sort(SortData* data, std::string* err) = 0;
};
struct ICloneSort {                   // sort.go:36-39: AB-experiment overrides ("sort.<name>" params object)
    virtual ~ICloneSort() = default;
    virtual std::shared_ptr<ISort> CloneWithConfig(const json::Value& params) = 0;
    virtual std::string GetSortName() const = 0;
};
class Registry {                      // sort.go:143-150: first registration wins; nil panics
public:
    bool RegisterSort(const std::string& name, std::shared_ptr<ISort> s, std::string* err);
    // registerSortWithSign (sort.go:204-207): what RegisterSortWithConfig uses — overwrites
    void RegisterSortWithConfig(const std::string& name, std::shared_ptr<ISort> s) { sorts_[name] = std::move(s); }
    std::shared_ptr<ISort> Get(const std::string& name);
private:
    std::map<std::string, std::shared_ptr<ISort>> sorts_;
};
}  // namespace sort

// ---- item ids (ingest.cpp) ------------------------------------------------------------------------------------------
// row ↔ module.ItemId: the ids back to back in one arena, an offset per row, an open-addressing table of row numbers
class IdDict {
public:
    void Reserve(uint64_t rows, uint64_t id_bytes);
    void Append(const char* id, size_t len);                 // the next row's id
    bool BuildIndex(std::string* err);                       // after the last Append; fails on a duplicate id
    uint64_t size() const { return off_.empty() ? 0 : off_.size() - 1; }
    std::string IdOf(uint64_t row) const { return std::string(arena_.data() + off_[row], (size_t)(off_[row + 1] - off_[row])); }
    bool RowOf(const char* id, size_t len, uint32_t* row) const;
private:
    static constexpr uint32_t kEmpty = 0xFFFFFFFFu;
    std::vector<char> arena_;
    std::vector<uint64_t> off_;
    std::unique_ptr<std::atomic<uint32_t>[]> slots_;
    uint64_t mask_ = 0;
};

// Readers (requests) share, the table generation change-over is exclusive and goes first once it waits
// (a plain pthread rwlock lets a steady stream of readers starve the writer)
class VersionLock {
public:
    struct Read {
        explicit Read(VersionLock& l) : l_(l) {
            std::unique_lock<std::mutex> g(l_.mu_);
            l_.cv_.wait(g, [&] { return !l_.writing_ && l_.writers_waiting_ == 0; });
            ++l_.readers_;
        }
        ~Read() {
            std::lock_guard<std::mutex> g(l_.mu_);
            if (--l_.readers_ == 0) l_.cv_.notify_all();
        }
        VersionLock& l_;
    };
    struct Write {
        explicit Write(VersionLock& l) : l_(l) {
            std::unique_lock<std::mutex> g(l_.mu_);
            ++l_.writers_waiting_;
            l_.cv_.wait(g, [&] { return !l_.writing_ && l_.readers_ == 0; });
            --l_.writers_waiting_;
            l_.writing_ = true;
        }
        ~Write() {
            std::lock_guard<std::mutex> g(l_.mu_);
            l_.writing_ = false;
            l_.cv_.notify_all();
        }
        VersionLock& l_;
    };
private:
    std::mutex mu_;
    std::condition_variable cv_;
    int readers_ = 0, writers_waiting_ = 0;
    bool writing_ = false;
};

// ---- the engine: GPU-backed plugins wired under the registries -------------------------------------
namespace feature { class Feature; }
class Engine {
public:
    ~Engine();
    static Engine* Create(const std::string& config_json, std::string* err);
    // Recommend (service/user_recommend.go:46-183 restricted to the hot path):
    // recall → UniqueFilter → rank → sort → items[:size]
    bool Recommend(const std::string& uid, int size, const std::string& scene,
                   std::vector<module::ItemPtr>* out, std::string* err);
    // the same with an AB experiment attached (ctx.ExperimentResult): params = the layer params object
    bool Recommend(const std::string& uid, int size, const std::string& scene, const json::Value& experiment_params,
                   std::vector<module::ItemPtr>* out, std::string* err);

    module::InMemoryVectorDao user_vectors;
    algorithm::AlgorithmFactory algorithms;
    recall::Registry recalls;
    sort::Registry sorts;
    recconf::RecommendConfig config;
    // FeatureService / UserFeatureService (service/feature/feature_service.go:77-131, user_feature_service.go): the transforms of
    // FeatureConfs[scene] run between the filter and the rank call, those of UserFeatureConfs[scene] before the recalls
    std::map<std::string, std::vector<std::shared_ptr<feature::Feature>>> sceneFeatures, sceneUserFeatures;

    pg_ctx* ctx = nullptr;
    pg_table* table = nullptr;
    pg_model* model = nullptr;
    std::map<std::string, pg_model*> named_models;      // multi-output rank algorithms: "<algo>/<output>" → DNN3 model
    // pairec_gpu.Algorithms[].Precision ("f32" | "bf16" | "bf16x3", default bf16): what a loader passing prec < 0 gets
    std::map<std::string, int> algo_precision;
    int default_dnn_precision = PG_PREC_BF16, default_fm2t_precision = PG_PREC_BF16;
    int PrecisionOf(const std::string& algo, int fallback) const {
        auto it = algo_precision.find(algo);
        return it == algo_precision.end() ? fallback : it->second;
    }
    pg_features* feats = nullptr;                       // item "context features" as device columns (EasyRec request flavour)
    std::map<std::string, std::vector<int32_t>> user_fields;   // uid → dictionary-encoded user categorical features
    pg_model* fm2t = nullptr;                           // FM + two-tower model: rank algorithm "fm2t", and the vector model of the online recall
    pg_table* item_emb = nullptr;                       // … and the item-tower outputs it searches
    uint64_t item_emb_rows = 0;
    // request coalescer (UserDefineConfs.pairec_gpu.Coalesce): per-request plug-in calls share table passes
    bool coalesce = false;
    uint32_t coalesce_wait_us = 0, coalesce_depth = 0;
    std::mutex co_mu;
    uint32_t coalesce_timeout_us = 0;                    // UserDefineConfs.pairec_gpu.TimeoutUs: deadline of every coalesced call
    static constexpr size_t kMaxSceneCoalescers = 4;
    std::map<uint32_t, pg_coalescer*> co_scene;         // by k (RecallCount): recalls, both rank algorithms, DPPSort
    std::map<uint32_t, pg_coalescer*> co_online;        // by k: OnlineVectorRecall over the item-embedding table
    std::vector<std::string> fm2t_columns;              // the FM model's item field columns (the "fm2t" algorithm's configuration)
    uint32_t fm2t_d_user = 0, fm2t_nuf = 0;
    std::map<std::string, std::pair<pg_coalescer*, pg_expr*>> co_page;   // page recalls: name → (coalescer, compiled RankScore)
    // Hologres recalls with a WhereClause: the admitted rows as a filtered view of the table (pg_table_view_create), built by the
    // first request of a table generation, with a coalescer of its own when Coalesce is on
    struct FilterView {
        pg_table* view = nullptr;
        pg_coalescer* co = nullptr;
        uint64_t generation = ~0ull;
        bool empty = false, failed = false;      // no row passes / the view could not be built (the per-call form serves)
    };
    std::map<std::string, FilterView> co_views;         // by recall name
    const FilterView* ViewFor(const recconf::RecallConfig& conf, uint32_t k, const pg_table* base = nullptr);   // base: the searched table (default: `table`)
    uint32_t item_emb_dim = 0;
    uint32_t dim_item_emb() const { return item_emb_dim; }
    void DropViewsLocked();
    pg_coalescer* SceneCoalescer(uint32_t k, std::string* err);
    pg_coalescer* OnlineCoalescer(uint32_t k, std::string* err);
    void DropCoalescers();
    pg_coalescer* PageCoalescer(const recconf::RecallConfig& conf, std::string* err);
    uint64_t table_rows = 0;
    uint32_t dim = 0;
    std::string id_prefix = "item_";          // without a dictionary: "<prefix><row>"
    // row ↔ item id: the dictionary of the table's current generation (ingest.cpp), else the prefix scheme
    std::shared_ptr<const IdDict> dict;
    bool RowOfId(const std::string& id, uint32_t* row) const;
    std::string IdOfRow(uint64_t row) const;
    // table generations: a request holds `version` shared from its first plug-in call to its last label lookup; a
    // commit swaps table and dictionary exclusively
    VersionLock version;
    std::atomic<uint64_t> generation{0};
    std::mutex ingest_mu;
    pg_table* staging = nullptr;
    std::unique_ptr<IdDict> staging_dict;
    uint64_t staging_filled = 0;
    // the row-keyed feature columns (WhereClause column, FM item fields) of the generation being loaded: they change over
    // with the rows and the ids, inside the same exclusive section
    pg_features* staging_feats = nullptr;
    bool IngestFeatureColumn(const std::string& name, const int32_t* values, uint64_t n, std::string* err);
    bool IngestBegin(std::string* err);
    bool IngestChunk(const char* ids, size_t ids_bytes, const float* rows, uint64_t n, std::string* err);
    bool IngestCommit(std::string* err);
    bool IngestFile(const std::string& path, std::string* err);       // the table's "Path" (UserDefineConfs.pairec_gpu.Table.Path)
};

namespace rank {
// EasyrecAlgoDataGenerator (service/rank/algo_data.go:172-306): the per-request host boxing that the device
// feature store (pg_features_*) replaces — kept here as the statement of its default-filling semantics.
// Values are JSON scalars; the "reflect type" of a value is int (integral number), float64 or string.
class EasyrecAlgoDataGenerator {
public:
    explicit EasyrecAlgoDataGenerator(const std::vector<std::string>& contextFeatures);   // :183-201
    void SetItemFeatures(const std::vector<std::string>& inputItemFeatures);              // :204-221
    void AddFeatures(const module::ItemPtr& item, const std::map<std::string, json::Value>& itemFeatures,
                     const std::map<std::string, json::Value>& userFeatures);             // :223-271
    // GeneratorAlgoData (:273-302): {"user_features":{..},"item_ids":[..],"context_features":{name:[..]},
    // "item_features":{name:[..]}} and the per-request lists are reset
    std::string GeneratorAlgoData();
    bool HasFeatures() const { return !requestItem_.empty(); }
private:
    struct Feature { std::string name; json::Value::Type type; bool is_int; json::Value Default() const; };
    std::vector<module::ItemPtr> requestItem_;
    std::map<std::string, std::vector<json::Value>> contextFeatures_, inputItemFeatureMap_;
    bool parseFeature_ = true, parseInputItemFeature_ = false, hasInputMap_ = false;
    std::vector<Feature> itemFeatures_, inputItemFeatures_;
    std::map<std::string, json::Value> userFeatures_;
};

// RankService.Rank (rank_service.go:102-372): batches of BatchCount, one algorithm.Run per batch
// and algo, scores written back with AddAlgoScore, Item.Score = RankScore expression.
bool Rank(Engine* e, module::User* user, std::vector<module::ItemPtr>& items, context::RecommendContext* ctx,
          std::string* err);
}

// ---- service/feature: normalizers + feature transforms ---------------------------------
// NewNormalizer (service/feature/normalizer.go:19-41) and the FeatureOp family (op.go:17-33, new_feature_op.go,
// delete_feature_op.go, batch_raw_feature_op.go) behind Feature.LoadFeatures (feature.go:17-41): host glue upstream of the
// rank call, mirrored so that a scene's FeatureLoadConfig reads the same here.  The two expression normalizers evaluate a
// stated SUBSET of their third-party languages (Knetic/govaluate v3.0.1-0.20171022003610 for "expression", expr-lang/expr
// v1.17.6 for "expr" — both go.mod dependencies absent from the reference tree): what is outside it is refused by name when the
// normalizer is built, never approximated.  feature.cpp states the subset.
namespace feature {
struct Normalizer {
    virtual ~Normalizer() = default;
    virtual json::Value Apply(const json::Value& value) = 0;     // Normalizer.Apply (normalizer.go:15-17)
    virtual const char* Kind() const = 0;                        // the Go type a caller switches on (new_feature_op.go:18-45)
};
// nullptr for a name the reference does not know (its NewNormalizer returns a nil interface) — and, with *err set, for an
// expression outside the subset
std::shared_ptr<Normalizer> NewNormalizer(const std::string& name, const std::string& expression, std::string* err);
// utils.GovaluateFunctions (utils/govaluate_functions.go:20-330) by name: the functions both expression languages share
bool CallFunction(const std::string& name, const std::vector<json::Value>& args, json::Value* out, std::string* err);
bool HasFunction(const std::string& name);
// the clock the time normalizers and timestamp() read: 0 = the system's (tests pin it)
void SetClockForTest(long long unix_millis);

using FeatureConfig = recconf::FeatureConfig;
class Feature {                             // feature.go:10-41 without the FeatureDao fetch (storage is out of scope)
public:
    bool LoadWithConfig(const std::vector<FeatureConfig>& features, std::string* err);
    void LoadFeatures(module::User* user, std::vector<module::ItemPtr>& items, context::RecommendContext* ctx);
private:
    struct Trans { FeatureConfig conf; std::string source; std::shared_ptr<Normalizer> normalizer; };
    std::vector<Trans> trans_;
};
// Item.StringProperty (module/item.go:101-122: a float64 prints as its truncated integer) and User.StringProperty
// (module/user.go:168-189: a float64 prints with strconv 'f', -1)
std::string ItemStringProperty(const module::Item& it, const std::string& key);
std::string UserStringProperty(const module::User& u, const std::string& key);
}  // namespace feature

}  // namespace pairec
