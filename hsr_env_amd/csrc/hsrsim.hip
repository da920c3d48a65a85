// libhsrsim.so - host side of the C-ABI declared in include/hsrsim.h (gfx950 only).
//
// Owns: model tables on the device (fp32), per-batch SoA state, one HIP stream per batch, and the launches: one persistent
// kernel per env-step (persist.h: all substeps, ctrl in, obs / reward / done out; a work queue when the batch has more tasks than
// resident workgroups), or - for models outside its lane maps, and as the cross-check of the tests - the per-substep chain
// k_kinematics -> k_cull + k_narrow -> k_solve_mf, optionally replayed from a captured hipGraph.
#include <hip/hip_runtime.h>
#include <math.h>
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/hsrsim.h"
#include "collide.h"
#include "model.h"
#include "solve_g.h"
#include "solve_mf.h"
#include "persist.h"
#include "cfg_consts.h"

static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, const char *detail = "") {
    snprintf(g_err, sizeof g_err, fmt, detail);
    return code;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(HSR_EDEVICE, "HIP error: %s", hipGetErrorString(e_)); } while (0)

extern "C" const char *hsr_last_error(void) { return g_err; }

// ------------------------------------------------------------------ blob parsing
struct BlobEntry { char name[32]; uint32_t dtype, ndim, shape[4]; uint64_t off, nbytes; };

struct hsr_model {
    std::vector<uint8_t> raw;
    std::map<std::string, const BlobEntry *> entries;
    const uint8_t *data = nullptr;
    std::string json;
    int sizes[16];
    double opt[16];
    std::vector<std::string> body_names, joint_names;
    std::vector<std::pair<int, int>> joint_qposadr;
    // host copies used to build device tables for each batch's device
    std::vector<float> ctrlrange, qpos0;

    const double *f64(const char *n, size_t *count = nullptr) const {
        auto it = entries.find(n);
        if (it == entries.end()) return nullptr;
        if (count) *count = it->second->nbytes / 8;
        return (const double *)(data + it->second->off);
    }
    const int *i32(const char *n, size_t *count = nullptr) const {
        auto it = entries.find(n);
        if (it == entries.end()) return nullptr;
        if (count) *count = it->second->nbytes / 4;
        return (const int *)(data + it->second->off);
    }
};

// minimal JSON helpers for the "names"/"meta" sidecar (flat lists of strings / int pairs)
static size_t json_find_key(const std::string &js, const char *key, size_t from = 0) {
    std::string k = std::string("\"") + key + "\":";
    return js.find(k, from);
}
static std::vector<std::string> json_string_list(const std::string &js, size_t pos) {
    std::vector<std::string> out;
    size_t lb = js.find('[', pos);
    if (lb == std::string::npos) return out;
    size_t i = lb + 1;
    while (i < js.size() && js[i] != ']') {
        if (js[i] == '"') {
            size_t j = js.find('"', i + 1);
            out.push_back(js.substr(i + 1, j - i - 1));
            i = j + 1;
        } else if (js.compare(i, 4, "null") == 0) { out.push_back(""); i += 4; }
        else i++;
    }
    return out;
}
static std::vector<std::pair<int, int>> json_pair_list(const std::string &js, size_t pos) {
    std::vector<std::pair<int, int>> out;
    size_t lb = js.find('[', pos);
    if (lb == std::string::npos) return out;
    size_t i = lb + 1;
    int depth = 1;
    std::vector<int> cur;
    while (i < js.size() && depth > 0) {
        char ch = js[i];
        if (ch == '[') { depth++; cur.clear(); i++; }
        else if (ch == ']') { depth--; if (depth == 1 && cur.size() == 2) out.push_back({cur[0], cur[1]}); i++; }
        else if ((ch >= '0' && ch <= '9') || ch == '-') { char *endp; long v = strtol(js.c_str() + i, &endp, 10); cur.push_back((int)v); i = endp - js.c_str(); }
        else i++;
    }
    return out;
}

extern "C" int hsr_model_load(const void *blob, size_t len, hsr_model **out) {
    if (!blob || !out) return fail(HSR_EINVAL, "null argument");
    if (len < 16 || memcmp(blob, "HSRM0001", 8) != 0) return fail(HSR_EBLOB, "not an HSRM0001 model blob");
    // the blob is untrusted input (hsr/mujoco_env.py:30-31: a bad model file is an IOError, never a crash): every length and offset
    // is checked against `len` before it is used
    const uint8_t *in = (const uint8_t *)blob;
    uint32_t n;
    memcpy(&n, in + 8, 4);
    if (n > 4096 || 16 + (uint64_t)n * sizeof(BlobEntry) + 8 > len) return fail(HSR_EBLOB, "truncated model blob (entry table)");
    uint64_t jl;
    memcpy(&jl, in + 16 + (size_t)n * sizeof(BlobEntry), 8);
    const uint64_t data_off = 16 + (uint64_t)n * sizeof(BlobEntry) + 8;
    if (jl > len - data_off || (jl & 7) != 0) return fail(HSR_EBLOB, "truncated model blob (names / meta)");
    const uint64_t data_len = len - data_off - jl;
    {
        const BlobEntry *ent0 = (const BlobEntry *)(in + 16);
        for (uint32_t i = 0; i < n; i++) {
            BlobEntry e;
            memcpy(&e, ent0 + i, sizeof e);
            if (e.off > data_len || e.nbytes > data_len - e.off) return fail(HSR_EBLOB, "model blob entry out of bounds");
            if ((e.off & 7) != 0) return fail(HSR_EBLOB, "model blob entry misaligned");
        }
    }
    hsr_model *m = new hsr_model();
    m->raw.assign(in, in + len);
    const uint8_t *raw = m->raw.data();
    const BlobEntry *ent = (const BlobEntry *)(raw + 16);
    const uint8_t *p = raw + 16 + (size_t)n * sizeof(BlobEntry);
    m->json.assign((const char *)p + 8, (size_t)jl);
    m->data = p + 8 + jl;
    for (uint32_t i = 0; i < n; i++) m->entries[std::string(ent[i].name, strnlen(ent[i].name, 32))] = &ent[i];
    size_t nsz = 0, nop = 0;
    const int *sz = m->i32("sizes", &nsz);
    const double *op = m->f64("opt", &nop);
    if (!sz || !op || nsz < 16 || nop < 16) { delete m; return fail(HSR_EBLOB, "blob lacks sizes/opt"); }
    memcpy(m->sizes, sz, sizeof m->sizes);
    memcpy(m->opt, op, sizeof m->opt);
    for (int i = 0; i < 16; i++) if (m->sizes[i] < 0 || m->sizes[i] > (1 << 20)) { delete m; return fail(HSR_EBLOB, "blob sizes out of range"); }
    {   // every table the host code and the kernels index by a model size must be at least that long, and every index table
        // must point inside the table it indexes: a corrupted file is refused here, not found by a kernel
        const int nq = m->sizes[HSR_NQ], nv = m->sizes[HSR_NV], nu = m->sizes[HSR_NU], nl = m->sizes[HSR_NLINK], nb = m->sizes[HSR_NBODY],
                  ng = m->sizes[HSR_NGEOM], np_ = m->sizes[HSR_NPAIR], nmv = m->sizes[HSR_NMESHVERT], nslot = m->sizes[HSR_NSLOT];
        struct Need { const char *name; int dtype; long long count; };
        const Need need[] = {
            {"qpos0", 0, nq}, {"link_parent", 1, nl}, {"link_pos", 0, 3LL * nl}, {"link_quat", 0, 4LL * nl}, {"link_dofadr", 1, nl}, {"link_dofnum", 1, nl},
            {"link_qposadr", 1, nl}, {"link_free", 1, nl}, {"link_mass", 0, nl}, {"link_com", 0, 3LL * nl}, {"link_inertia", 0, 6LL * nl}, {"link_dofmask", 1, nl},
            {"dof_link", 1, nv}, {"dof_type", 1, nv}, {"dof_axis", 0, 3LL * nv}, {"dof_pos", 0, 3LL * nv}, {"dof_parent", 1, nv}, {"dof_damping", 0, nv},
            {"dof_qposadr", 1, nv}, {"dof_invweight0", 0, nv}, {"dof_limited", 1, nv}, {"dof_range", 0, 2LL * nv}, {"dof_solref", 0, 2LL * nv}, {"dof_solimp", 0, 5LL * nv},
            {"body_link", 1, nb}, {"body_pos", 0, 3LL * nb}, {"body_quat", 0, 4LL * nb}, {"body_mocap", 1, nb},
            {"geom_type", 1, ng}, {"geom_link", 1, ng}, {"geom_pos", 0, 3LL * ng}, {"geom_quat", 0, 4LL * ng}, {"geom_size", 0, 3LL * ng}, {"geom_rbound", 0, ng},
            {"geom_meshadr", 1, ng}, {"geom_meshnum", 1, ng}, {"geom_invweight", 0, 2LL * ng}, {"geom_aabb", 0, 6LL * ng}, {"mesh_vert", 0, 3LL * nmv},
            {"pair_geom1", 1, np_}, {"pair_geom2", 1, np_}, {"pair_fn", 1, np_}, {"pair_condim", 1, np_}, {"pair_slot", 1, np_ + 1LL}, {"pair_friction", 0, 5LL * np_},
            {"pair_solref", 0, 2LL * np_}, {"pair_solimp", 0, 5LL * np_},
            {"act_dof", 1, nu}, {"act_gear", 0, nu}, {"act_kp", 0, nu}, {"act_ctrlrange", 0, 2LL * nu}, {"act_forcerange", 0, 2LL * nu}};
        for (const Need &nd : need) {
            auto it = m->entries.find(nd.name);
            if (it == m->entries.end()) { delete m; return fail(HSR_EBLOB, "blob entry '%s' missing", nd.name); }
            if ((int)it->second->dtype != nd.dtype || (long long)(it->second->nbytes / (nd.dtype == 0 ? 8 : 4)) < nd.count) { delete m; return fail(HSR_EBLOB, "blob entry '%s' shorter than the model sizes say", nd.name); }
        }
        auto in_range = [&](const char *name, int cnt, int lo, int hi) {       // all of the first cnt values in [lo, hi)
            const int *v = m->i32(name);
            for (int i = 0; i < cnt; i++) if (v[i] < lo || v[i] >= hi) return false;
            return true;
        };
        bool ok = nl >= 1 && in_range("link_parent", nl, 0, nl) && in_range("dof_link", nv, 0, nl) && in_range("dof_parent", nv, -1, nv) && in_range("dof_qposadr", nv, 0, nq > 0 ? nq : 1)
                  && in_range("body_link", nb, 0, nl) && in_range("geom_link", ng, 0, nl) && in_range("pair_geom1", np_, 0, ng) && in_range("pair_geom2", np_, 0, ng)
                  && in_range("pair_fn", np_, 0, 4) && in_range("pair_slot", np_ + 1, 0, nslot + 1) && in_range("act_dof", nu, 0, nv > 0 ? nv : 1)
                  && in_range("link_dofadr", nl, -1, nv + 1) && in_range("link_dofnum", nl, 0, nv + 1) && in_range("link_qposadr", nl, -1, nq + 1);
        if (ok) {
            const int *ma = m->i32("geom_meshadr"), *mn = m->i32("geom_meshnum"), *gt = m->i32("geom_type");
            for (int g = 0; g < ng; g++) if (gt[g] == GEOM_MESH && (ma[g] < 0 || mn[g] < 0 || (long long)ma[g] + mn[g] > nmv)) ok = false;
        }
        if (!ok) { delete m; return fail(HSR_EBLOB, "blob index table out of range"); }
    }
    size_t np = json_find_key(m->json, "names");
    if (np != std::string::npos) {
        size_t bp = json_find_key(m->json, "body", np), jp = json_find_key(m->json, "joint", np);
        if (bp != std::string::npos) m->body_names = json_string_list(m->json, bp);
        if (jp != std::string::npos) m->joint_names = json_string_list(m->json, jp);
    }
    size_t qp = json_find_key(m->json, "joint_qposadr");
    if (qp != std::string::npos) m->joint_qposadr = json_pair_list(m->json, qp);
    const int nu = m->sizes[HSR_NU], nq = m->sizes[HSR_NQ];
    size_t ncr = 0, nq0 = 0;
    const double *cr = m->f64("act_ctrlrange", &ncr), *q0 = m->f64("qpos0", &nq0);
    if ((nu > 0 && (!cr || ncr < (size_t)nu * 2)) || (nq > 0 && (!q0 || nq0 < (size_t)nq))) { delete m; return fail(HSR_EBLOB, "blob lacks act_ctrlrange / qpos0"); }
    m->ctrlrange.resize((size_t)nu * 2);
    for (int i = 0; i < nu * 2; i++) m->ctrlrange[i] = (float)cr[i];
    m->qpos0.resize(nq);
    for (int i = 0; i < nq; i++) m->qpos0[i] = (float)q0[i];
    *out = m;
    return HSR_OK;
}
extern "C" void hsr_model_destroy(hsr_model *m) { delete m; }
extern "C" int hsr_model_size(const hsr_model *m, int which) { return (m && which >= 0 && which < 16) ? m->sizes[which] : HSR_EINVAL; }
extern "C" double hsr_model_timestep(const hsr_model *m) { return m ? m->opt[0] : 0.0; }
extern "C" int hsr_model_ctrlrange(const hsr_model *m, float *out) {
    if (!m || !out) return fail(HSR_EINVAL, "null argument");
    memcpy(out, m->ctrlrange.data(), m->ctrlrange.size() * sizeof(float)); return HSR_OK;
}
extern "C" int hsr_model_qpos0(const hsr_model *m, float *out) {
    if (!m || !out) return fail(HSR_EINVAL, "null argument");
    memcpy(out, m->qpos0.data(), m->qpos0.size() * sizeof(float)); return HSR_OK;
}
extern "C" int hsr_model_body_id(const hsr_model *m, const char *name) {
    if (!m || !name) return fail(HSR_EINVAL, "null argument");
    for (size_t i = 0; i < m->body_names.size(); i++) if (m->body_names[i] == name) return (int)i;
    return fail(HSR_ENAME, "unknown body '%s'", name);
}
extern "C" int hsr_model_joint_qpos_addr(const hsr_model *m, const char *name, int *start, int *end) {
    if (!m || !name || !start || !end) return fail(HSR_EINVAL, "null argument");
    for (size_t i = 0; i < m->joint_names.size() && i < m->joint_qposadr.size(); i++)
        if (m->joint_names[i] == name) { *start = m->joint_qposadr[i].first; *end = m->joint_qposadr[i].first + m->joint_qposadr[i].second; return HSR_OK; }
    return fail(HSR_ENAME, "unknown joint '%s'", name);
}

// ------------------------------------------------------------------ batch
struct GraphKey { int nsub, goal_body; float geofence; bool operator<(const GraphKey &o) const { return std::tie(nsub, goal_body, geofence) < std::tie(o.nsub, o.goal_body, o.geofence); } };

struct hsr_batch {
    const hsr_model *model = nullptr;
    int N = 0, device = 0;
    hipStream_t stream = nullptr;
    DevModel dm{};
    DevModel *d_dm = nullptr;       // device copy for kernels that take the model by pointer
    DevState ds{};
    std::vector<void *> allocs;
    float *d_qpos0 = nullptr;      // model qpos0 on the device
    float *d_stage = nullptr;      // staging for host-pointer API: max(N*(nq+nv), ...) floats
    size_t stage_floats = 0;
    uint8_t *d_stage_u8 = nullptr;
    int32_t *d_stage_i32 = nullptr;
    int narrow_blocks = 2048;      // persistent-style grid of k_narrow (HSR_NARROW_BLOCKS overrides)
    int pairs_per_wave = 4;        // k_collide: pairs walked by one wave (HSR_PPW overrides)
    int group = 16;                // lanes per env of the cooperative solver
    size_t mf_lds_bytes = 0, persist_lds_bytes = 0;
    bool persist = false;          // whole env-step in one persistent kernel (k_env_step_mf); HSR_PERSIST=0 disables
    bool persist_ok = false;       // the model fits the persistent kernel (lane maps, LDS, kinematic structure): set once at creation
    bool use_graph = true, profiling = false, debug_store = false;
    bool persist_tg = false;       // persistent kernel instance that reads its pair / geom tables from global memory (LDS budget)
    bool mpr_warm = true;          // penetrating convex pairs start MPR from the portal of their previous substep (HSR_MPR_WARM=0 / hsr_batch_set_mpr_warm turn it off)
    int test_hooks = 0;            // hsr_batch_set_debug bits 1.. : force rarely taken solver branches (tests only)
    bool schedule = true;          // re-pack the envs over the waves of the persistent kernel before every launch (HSR_SCHEDULE=0 / hsr_batch_set_schedule turn it off)
    int *d_slot_env = nullptr;
    std::map<GraphKey, hipGraphExec_t> graphs;
    float last_total_ms = 0, last_kernel_ms[3] = {0, 0, 0};
    int last_launches[3] = {0, 0, 0};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> kev;
    int slots = 0;                 // workgroups of the persistent kernel the GPU holds at once (occupancy x compute units)
    int queue = -1;                // work queue of the persistent kernel: -1 = automatic (on when there are more tasks than slots), 0 / 1 forced (HSR_QUEUE)
    int queue_chunk = 20;          // substeps per round of the work queue (HSR_QUEUE_CHUNK)
    bool queue_chunk_set = false;  // ... chosen by the caller (environment / hsr_batch_set_queue): no automatic choice then
    int solo_servers = 0;          // workgroups of a queued launch that run hard envs alone (persist.h; hsr_batch_set_solo / HSR_SOLO); 0 = off
    float solo_trips = 3.5f;       // hand-over threshold: Newton iterations per substep over a round
    bool solo_ok = false;          // the chosen kernel instance has the server path
    bool kernel_log = false;       // hsr_batch_set_profiling(b, 2): an event pair around every launch of the persistent kernel, no synchronisation
    std::vector<std::pair<hipEvent_t, hipEvent_t>> klog;
};

// global copies of the two constant LDS tables of the persistent kernel (same packing: kin2.h)
__global__ void k_build_tables(DevModel m, DevState s) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m.ngeom) { geom_consts_store(m.geom_rec + 32 * i, s.geom_c + 8 * i); if (m.nldsv > 0 && m.geom_ldsv[i] >= 0) hull_lds_patch(s.geom_c + 8 * i, m.geom_ldsv[i]); }
    if (i < ((m.npair_pad + 7) & ~7)) {
        unsigned pk = 0;
        if (i < m.npair) {
            const float4 a = reinterpret_cast<const float4 *>(m.pair_geo)[2 * i], b = reinterpret_cast<const float4 *>(m.pair_geo)[2 * i + 1];
            const int code = (int)a.x;
            pk = pair_pack(code & 255, (int)a.y, (int)b.x, (code >> 8) ? a.w : a.z + a.w);
        }
        s.pair_pack[i] = pk;
    }
}

// The persistent kernel instantiations.  Every reference configuration has an instance with ALL scalar model fields at compile time
// (cfg_consts.h, generated from the committed blobs; chosen only when the loaded model matches the generated row value for value -
// HSR_NO_CONST=1 never chooses them); any other model runs a generic instance (lanes per env, bound on nv).
typedef void (*persist_fn)(const DevModel *, DevState, int, int, float, int, StepIO);
static void quat2mat_h(const double *q, float *mt);
// row + 1 of the constant instance whose compile-time tree tables (cfg_consts.h: Kin3_<cfg>) equal the loaded model's, 0 if none: the order of
// the entries is the one tools/gen_cfg_consts.py writes
static int kin3_matching_row(const hsr_model *m) {
    const int nv = m->sizes[1], nlink = m->sizes[3];
    std::vector<int> iv = {nlink, nv};
    for (const char *n : {"link_parent", "link_free", "link_dofadr", "link_dofnum", "link_qposadr", "dof_type", "dof_qposadr", "dof_link"}) {
        size_t cnt = 0; const int *p = m->i32(n, &cnt);
        if (!p) return 0;
        iv.insert(iv.end(), p, p + cnt);
    }
    std::vector<float> fv;
    auto addf = [&](const char *n) { size_t cnt = 0; const double *p = m->f64(n, &cnt); if (!p) return false; for (size_t i = 0; i < cnt; i++) fv.push_back((float)p[i]); return true; };
    if (!addf("link_pos")) return 0;
    { size_t cnt = 0; const double *q = m->f64("link_quat", &cnt); if (!q) return 0; for (size_t i = 0; i < cnt / 4; i++) { float mt[9]; quat2mat_h(q + 4 * i, mt); fv.insert(fv.end(), mt, mt + 9); } }
    if (!addf("link_com") || !addf("link_inertia") || !addf("link_mass") || !addf("dof_axis") || !addf("dof_pos")) return 0;
    for (size_t r = 0; r < sizeof kKin3Checks / sizeof kKin3Checks[0]; r++) {
        const Kin3Check &k = kKin3Checks[r];
        if (k.i && k.ni == (int)iv.size() && k.nf == (int)fv.size() && memcmp(k.i, iv.data(), iv.size() * sizeof(int)) == 0 && memcmp(k.f, fv.data(), fv.size() * sizeof(float)) == 0) return (int)r + 1;
    }
    return 0;
}
static int cfg_const_row(const DevModel &d) {
    const char *nc = getenv("HSR_NO_CONST");
    if (nc && strcmp(nc, "0") != 0) return -1;
    int iv[sizeof kCfgConsts[0].i / sizeof(int)]; float fv[sizeof kCfgConsts[0].f / sizeof(float)];
    cfg_const_values(d, iv, fv);
    for (size_t r = 0; r < sizeof kCfgConsts / sizeof kCfgConsts[0]; r++)
        if (memcmp(iv, kCfgConsts[r].i, sizeof iv) == 0 && memcmp(fv, kCfgConsts[r].f, sizeof fv) == 0) {
            // an instance compiled with its kinematic tree (kin3.h) serves only a model whose tree is that one, value for value (hsr_batch_create: kin3_match)
            if (kKin3Checks[r].i && d.kin3_match != (int)r + 1) return -1;
            return (int)r;
        }
    return -1;
}
enum { QUEUE_ROUNDS = 64 };
// the instance with the solo-server path (persist.h SV) of the configurations that have one, else NULL
static persist_fn persist_kernel_sv(const DevModel &d, int group, bool tg) {
    const int row = cfg_const_row(d);
    if (group != 16) return nullptr;
#ifdef HSR_DEV_CFG3
    return (!tg && row == 2) ? k_env_step_mf<16, 13, true, 7, false, DevModel_cfg3, true> : nullptr;
#else
    if (tg) return row == 4 ? k_env_step_mf<16, 13, true, -1, true, DevModel_cupboard, true> : nullptr;
    switch (row) {
    case 1: return k_env_step_mf<16, 8, true, 0, false, DevModel_cfg2, true>;
    case 2: return k_env_step_mf<16, 13, true, 7, false, DevModel_cfg3, true>;
    case 4: return k_env_step_mf<16, 13, true, -1, false, DevModel_cupboard, true>;
    default: return nullptr;
    }
#endif
}
static persist_fn persist_kernel(const DevModel &d, int group, bool tg = false) {
    const int row = cfg_const_row(d);
    const int nv = d.nv;
#ifdef HSR_DEV_CFG3
    // development builds (tools/build_variants.py): only the cfg3 instance is compiled - a sixth of the build time
    return (!tg && row == 2) ? k_env_step_mf<16, 13, true, 7, false, DevModel_cfg3> : nullptr;
#else
    if (tg) {      // pair / geom tables in global memory (LDS budget: 8 workgroups per CU)
        if (row == 4) return k_env_step_mf<16, 13, true, -1, true, DevModel_cupboard>;      // 274 candidate pairs
        if (row == 3) return k_env_step_mf<32, 25, true, 7, true, DevModel_cfg4>;           // 124 constraint rows per env (compiler.py: eff_njmax)
        return nullptr;
    }
    switch (row) {
    case 0: return k_env_step_mf<16, 2, true, 0, false, DevModel_cfg1>;            // two orthogonal slides
    case 1: return k_env_step_mf<16, 8, true, 0, false, DevModel_cfg2>;            // the slides + one block
    case 2: return k_env_step_mf<16, 13, true, 7, false, DevModel_cfg3>;           // arm + block
    case 3: return k_env_step_mf<32, 25, true, 7, false, DevModel_cfg4>;           // arm + three blocks
    case 4: return k_env_step_mf<16, 13, true, -1, false, DevModel_cupboard>;      // cupboard with its tables in LDS (HSR_TABLES_GLOBAL=0: 7 workgroups per CU)
    default: break;
    }
    if (group == 16) {
        if (nv == 13) return k_env_step_mf<16, 13, true>;                            // ndense at run time
        return k_env_step_mf<16, 16, false>;
    }
    return k_env_step_mf<32, 32, false>;
#endif
}

template <typename T>
static int dalloc(hsr_batch *b, T **p, size_t count) {
    void *q = nullptr;
    HIPCHK(hipMalloc(&q, (count ? count : 1) * sizeof(T)));
    HIPCHK(hipMemset(q, 0, (count ? count : 1) * sizeof(T)));
    b->allocs.push_back(q);
    *p = (T *)q;
    return HSR_OK;
}
static int upload_f(hsr_batch *b, const float **dst, const hsr_model *m, const char *name) {
    size_t cnt = 0;
    const double *src = m->f64(name, &cnt);
    if (!src) return fail(HSR_EBLOB, "blob entry '%s' missing", name);
    std::vector<float> tmp(cnt ? cnt : 1, 0.f);
    for (size_t i = 0; i < cnt; i++) tmp[i] = (float)src[i];
    float *d;
    int rc = dalloc(b, &d, tmp.size());
    if (rc) return rc;
    HIPCHK(hipMemcpy(d, tmp.data(), tmp.size() * sizeof(float), hipMemcpyHostToDevice));
    *dst = d;
    return HSR_OK;
}
static int upload_i(hsr_batch *b, const int **dst, const hsr_model *m, const char *name) {
    size_t cnt = 0;
    const int *src = m->i32(name, &cnt);
    if (!src) return fail(HSR_EBLOB, "blob entry '%s' missing", name);
    int *d;
    int rc = dalloc(b, &d, cnt + 16);          // zero padding: the solver reads pair_slot in rows of eight (solve_body.inc, E2)
    if (rc) return rc;
    if (cnt) HIPCHK(hipMemcpy(d, src, cnt * sizeof(int), hipMemcpyHostToDevice));
    *dst = d;
    return HSR_OK;
}

static void quat2mat_h(const double *q, float *mt) {
    double n = sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
    double w = q[0]/n, x = q[1]/n, y = q[2]/n, z = q[3]/n;
    mt[0] = (float)(1 - 2*(y*y + z*z)); mt[1] = (float)(2*(x*y - w*z)); mt[2] = (float)(2*(x*z + w*y));
    mt[3] = (float)(2*(x*y + w*z)); mt[4] = (float)(1 - 2*(x*x + z*z)); mt[5] = (float)(2*(y*z - w*x));
    mt[6] = (float)(2*(x*z - w*y)); mt[7] = (float)(2*(y*z + w*x)); mt[8] = (float)(1 - 2*(x*x + y*y));
}
static int upload_mats(hsr_batch *b, const float **dst, const hsr_model *m, const char *quat_name) {
    size_t cnt = 0;
    const double *q = m->f64(quat_name, &cnt);
    if (!q) return fail(HSR_EBLOB, "blob entry '%s' missing", quat_name);
    size_t n = cnt / 4;
    std::vector<float> tmp(n * 9 + 1);
    for (size_t i = 0; i < n; i++) quat2mat_h(q + 4 * i, tmp.data() + 9 * i);
    float *d;
    int rc = dalloc(b, &d, tmp.size());
    if (rc) return rc;
    HIPCHK(hipMemcpy(d, tmp.data(), tmp.size() * sizeof(float), hipMemcpyHostToDevice));
    *dst = d;
    return HSR_OK;
}

// ------------------------------------------------------------------ small layout / IO kernels
__global__ void k_aos_to_soa(float *dst, const float *src, int rows, int N) {   // src [N,rows] -> dst [rows][N]
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * N) return;
    int r = (int)(i / N), e = (int)(i % N);
    dst[i] = src[(size_t)e * rows + r];
}
__global__ void k_soa_to_aos(float *dst, const float *src, int rows, int N, int dst_stride, int dst_off) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * N) return;
    int r = (int)(i / N), e = (int)(i % N);
    dst[(size_t)e * dst_stride + dst_off + r] = src[i];
}
__global__ void k_begin_step(DevState s, const float *ctrl_in, int nu) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= s.N) return;
    for (int a = 0; a < nu; a++) s.ctrl[(size_t)a * s.N + e] = ctrl_in[(size_t)e * nu + a];
    s.done[e] = 0;
    s.nsteps[e] = 0;
}
__global__ void k_end_step(DevState s, float *reward, uint8_t *done, int32_t *nsteps) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= s.N) return;
    if (reward) reward[e] = s.done[e] ? 1.f : 0.f;
    if (done) done[e] = (uint8_t)(s.done[e] != 0);
    if (nsteps) nsteps[e] = s.nsteps[e];
}
__global__ void k_clear_done(DevState s) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < s.N) s.done[e] = 0;
}
// park: mask == NULL means "the envs whose done flag is set", and the envs that are not reset are left parked (done = 1) for the
// forward pass of the reset ones (hsr_batch_reset_dev clears the flags after it)
__global__ void k_reset(DevModel m, DevState s, const uint8_t *mask, const float *qpos0_env, const float *qpos0_model, const float *mocap, int park) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= s.N) return;
    const bool sel = mask ? mask[e] != 0 : (park ? s.done[e] != 0 : true);
    s.done[e] = (park && !sel) ? 1 : 0;
    if (!sel) return;
    const int N = s.N;
    for (int i = 0; i < m.nq; i++) s.qpos[(size_t)i * N + e] = qpos0_env ? qpos0_env[(size_t)e * m.nq + i] : qpos0_model[i];
    for (int i = 0; i < m.nv; i++) { s.qvel[(size_t)i * N + e] = 0; s.warm[(size_t)i * N + e] = 0; s.qacc[(size_t)i * N + e] = 0; }
    for (int i = 0; i < m.nu; i++) s.ctrl[(size_t)i * N + e] = 0;
    for (int k = 0; k < 3; k++) s.mocap[(size_t)k * N + e] = mocap ? mocap[(size_t)e * 3 + k] : 0.f;
    s.time[e] = 0; s.bad[e] = 0; s.nsteps[e] = 0;
    for (int p = 0; p < m.npair; p++) {      // the geoms jumped: no separation margin is left, and no portal of the previous substep (margin row -1: rows 0-2 hold its vertex ids)
        float *mg = s.sepax + (size_t)(4 * p + 3) * N + e;
        if (*mg < 0.f) { mg[-(ptrdiff_t)N] = 0.f; mg[-2 * (ptrdiff_t)N] = 0.f; mg[-3 * (ptrdiff_t)N] = 0.f; }
        *mg = 0.f;
    }
}
__global__ void k_body_xpos(DevModel m, DevState s, int body, float *out) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= s.N) return;
    const int N = s.N;
    v3 p;
    if (m.body_mocap[body]) p = mk3(s.mocap[e], s.mocap[N + e], s.mocap[2 * N + e]);
    else {
        const int l = m.body_link[body];
        View xpos{s.xpos + e, N}, xmat{s.xmat + e, N};
        p = xpos.get3(l) + mulmv(xmat.getm(l), ld3(m.body_pos, body));
    }
    out[3 * e] = p.x; out[3 * e + 1] = p.y; out[3 * e + 2] = p.z;
}
// The reference's 'openai' observation (hsr/env.py:72-110, after gym's FetchEnv) with its evident intent restored
// (SURVEY.md 8a-5 lists the defects): 25 floats per env =
//   grip_pos 3 | object_pos 3 | object_rel_pos 3 | gripper_state 2 (finger joint qpos) | object_rot 3 (mat2euler) |
//   object_velp 3 ((v_obj - v_grip) dt) | object_velr 3 (w_obj dt) | grip_velp 3 (v_grip dt) | gripper_vel 2 (dt/2 finger qvel)
// body positions / velocities are those of the last forward pass (sim.data.xpos / cvel after mj_step), joint values the
// current ones, dt = nsubsteps * timestep with nsubsteps = 1.
__device__ __forceinline__ void body_pose_vel(const DevModel &m, const DevState &s, int body, int e, v3 &p, v3 &v, v3 &w, m3 &R) {
    const int N = s.N, l = m.body_link[body], nl = m.nlink;
    const View xpos{s.xpos + e, N}, xmat{s.xmat + e, N};
    const m3 Rl = xmat.getm(l);
    const v3 off = mulmv(Rl, ld3(m.body_pos, body));
    p = xpos.get3(l) + off;
    R = mulmm(Rl, ldm(m.body_mat, body));
    w = mk3(s.lvel[(size_t)(3 * l) * N + e], s.lvel[(size_t)(3 * l + 1) * N + e], s.lvel[(size_t)(3 * l + 2) * N + e]);
    const v3 vo = mk3(s.lvel[(size_t)(3 * nl + 3 * l) * N + e], s.lvel[(size_t)(3 * nl + 3 * l + 1) * N + e], s.lvel[(size_t)(3 * nl + 3 * l + 2) * N + e]);
    v = vo + cross(w, off);
}
__global__ void k_obs_openai(DevModel m, DevState s, int body_l, int body_r, int body_obj, int qadr_l, int qadr_r, int dadr_l, int dadr_r,
                             float dt, float *out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= s.N) return;
    const int N = s.N;
    v3 pl, vl, wl, pr, vr, wr, po, vo, wo;
    m3 Rl, Rr, Ro;
    body_pose_vel(m, s, body_l, e, pl, vl, wl, Rl);
    body_pose_vel(m, s, body_r, e, pr, vr, wr, Rr);
    body_pose_vel(m, s, body_obj, e, po, vo, wo, Ro);
    const v3 grip = (pl + pr) * 0.5f, gvel = (vl + vr) * (0.5f * dt);
    const v3 rel = po - grip, ovel = vo * dt - gvel, orot = wo * dt;
    // mat2euler (hsr/env.py:256-272)
    const float cy = sqrtf(Ro.a[8] * Ro.a[8] + Ro.a[5] * Ro.a[5]);
    const bool cond = cy > 4.f * 2.220446049250313e-16f;
    const float ez = cond ? -atan2f(Ro.a[1], Ro.a[0]) : -atan2f(-Ro.a[3], Ro.a[4]);
    const float ey = -atan2f(-Ro.a[2], cy);
    const float ex = cond ? -atan2f(Ro.a[5], Ro.a[8]) : 0.f;
    float *o = out + (size_t)25 * e;
    o[0] = grip.x; o[1] = grip.y; o[2] = grip.z; o[3] = po.x; o[4] = po.y; o[5] = po.z; o[6] = rel.x; o[7] = rel.y; o[8] = rel.z;
    o[9] = s.qpos[(size_t)qadr_l * N + e]; o[10] = s.qpos[(size_t)qadr_r * N + e];
    o[11] = ex; o[12] = ey; o[13] = ez;
    o[14] = ovel.x; o[15] = ovel.y; o[16] = ovel.z; o[17] = orot.x; o[18] = orot.y; o[19] = orot.z;
    o[20] = gvel.x; o[21] = gvel.y; o[22] = gvel.z;
    o[23] = 0.5f * dt * s.qvel[(size_t)dadr_l * N + e]; o[24] = 0.5f * dt * s.qvel[(size_t)dadr_r * N + e];
}
__global__ void k_i32_to_f32(float *dst, const int *src, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}
__global__ void k_contacts_out(DevModel m, DevState s, float *out) {   // [N, nslot, 7]; empty slot: dist = +1
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= s.N) return;
    const int N = s.N;
    for (int p = 0; p < m.npair; p++) {
        const int cnt = s.ncon_pair[(size_t)e * m.npair_pad + p];
        for (int slot = m.pair_slot[p]; slot < m.pair_slot[p + 1]; slot++) {
            float *o = out + ((size_t)e * m.nslot + slot) * 7;
            const bool used = slot - m.pair_slot[p] < cnt;
            for (int k = 0; k < 7; k++) o[k] = used ? s.con[((size_t)e * m.nslot + slot) * 8 + k] : (k == 6 ? 1.f : 0.f);
        }
    }
}
__global__ void k_expand_M(DevState s, float *out, int nv) {   // packed [nM][N] -> [N,nv,nv]
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= s.N) return;
    for (int i = 0; i < nv; i++) for (int j = 0; j <= i; j++) {
        const float v = s.M[(size_t)(i * (i + 1) / 2 + j) * s.N + e];
        out[((size_t)e * nv + i) * nv + j] = v; out[((size_t)e * nv + j) * nv + i] = v;
    }
}

static inline dim3 grid1(size_t n, int t = 256) { return dim3((unsigned)((n + t - 1) / t)); }

static int batch_init(hsr_batch *b, const hsr_model *m, int n_envs);
extern "C" int hsr_batch_create(const hsr_model *m, int n_envs, int device_id, hsr_batch **out) {
    if (!m || !out || n_envs <= 0) return fail(HSR_EINVAL, "bad arguments to hsr_batch_create");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(HSR_EDEVICE, "no HIP device available%s");
    if (device_id < 0 || device_id >= ndev) return fail(HSR_EINVAL, "device id out of range");
    HIPCHK(hipSetDevice(device_id));
    hsr_batch *b = new hsr_batch();
    b->model = m; b->N = n_envs; b->device = device_id;
    const int rc = batch_init(b, m, n_envs);
    if (rc) { hsr_batch_destroy(b); return rc; }       // frees the stream, the events and every allocation made so far
    *out = b;
    return HSR_OK;
}
static int batch_init(hsr_batch *b, const hsr_model *m, int n_envs) {
    HIPCHK(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    DevModel &d = b->dm;
    const int *sz = m->sizes;
    d.nq = sz[HSR_NQ]; d.nv = sz[HSR_NV]; d.nu = sz[HSR_NU]; d.nlink = sz[HSR_NLINK]; d.nbody = sz[HSR_NBODY];
    d.ngeom = sz[HSR_NGEOM]; d.npair = sz[HSR_NPAIR]; d.nslot = sz[HSR_NSLOT]; d.nconmax = sz[HSR_NCONMAX]; d.njmax = sz[HSR_NJMAX];
    if (d.npair > 384) return fail(HSR_EINVAL, "more than 384 candidate geom pairs");
    if (d.ngeom > 255) return fail(HSR_EINVAL, "more than 255 geoms");
    d.nM = d.nv * (d.nv + 1) / 2;
    d.ndense = sz[13];
    d.timestep = (float)m->opt[0]; d.impratio = (float)m->opt[1]; d.gravz = (float)m->opt[2]; d.tolerance = (float)m->opt[3];
    d.iterations = (int)m->opt[4]; d.ls_iterations = (int)m->opt[5]; d.ls_tolerance = (float)m->opt[6];
    d.mpr_tolerance = (float)m->opt[7]; d.mpr_iterations = (int)m->opt[8]; d.meaninertia = (float)m->opt[9];
    int rc = 0;
#define UI(f) if ((rc = upload_i(b, &d.f, m, #f))) { return rc; }
#define UF(f) if ((rc = upload_f(b, &d.f, m, #f))) { return rc; }
    UI(link_parent) UI(link_dofadr) UI(link_dofnum) UI(link_qposadr) UI(link_free)
    UF(link_pos) UF(link_mass) UF(link_com) UF(link_inertia) UI(link_dofmask)
    UI(dof_link) UI(dof_type) UI(dof_parent) UI(dof_qposadr) UI(dof_limited)
    UF(dof_axis) UF(dof_pos) UF(dof_damping) UF(dof_invweight0) UF(dof_range) UF(dof_solref) UF(dof_solimp)
    UI(body_link) UI(body_mocap) UF(body_pos)
    UI(geom_type) UI(geom_link) UI(geom_meshadr) UI(geom_meshnum)
    UF(geom_pos) UF(geom_size) UF(geom_rbound) UF(geom_invweight) UF(mesh_vert) UF(geom_aabb)
    UI(pair_geom1) UI(pair_geom2) UI(pair_fn) UI(pair_condim) UI(pair_slot)
    UF(pair_friction) UF(pair_solref) UF(pair_solimp)
    UI(act_dof) UF(act_gear) UF(act_kp) UF(act_ctrlrange) UF(act_forcerange)
#undef UI
#undef UF
    {   // tree depth of every link: the persistent kernel walks the tree level by level with lane = link
        const int *lp = m->i32("link_parent");
        std::vector<int> dep(std::max(d.nlink, 1), 0);
        d.maxdepth = 0;
        for (int l = 1; l < d.nlink; l++) { dep[l] = dep[lp[l]] + 1; d.maxdepth = std::max(d.maxdepth, dep[l]); }
        int *dd; if ((rc = dalloc(b, &dd, dep.size()))) return rc;
        HIPCHK(hipMemcpy(dd, dep.data(), dep.size() * sizeof(int), hipMemcpyHostToDevice));
        d.link_depth = dd;
    }
    {   // hull vertices as float4
        size_t cnt = 0;
        const double *mv = m->f64("mesh_vert", &cnt);
        const size_t nvt = cnt / 3;
        std::vector<float> v4((nvt + 1) * 4, 0.f);
        for (size_t i = 0; i < nvt; i++) for (int k = 0; k < 3; k++) v4[4 * i + k] = (float)mv[3 * i + k];
        float *dv; if ((rc = dalloc(b, &dv, v4.size()))) return rc;
        HIPCHK(hipMemcpy(dv, v4.data(), v4.size() * sizeof(float), hipMemcpyHostToDevice));
        d.mesh_vert4 = reinterpret_cast<const float4 *>(dv);
        const int *mn = m->i32("geom_meshnum");
        for (int g = 0; g < d.ngeom; g++) if (mn[g] > 256) return fail(HSR_EINVAL, "mesh hull with more than 256 vertices");
    }
    {   // static geoms (world link) form a prefix of the geom list in every compiled model; anything else counts as moving
        const int *gl = m->i32("geom_link");
        d.nstatic_geom = 0;
        while (d.nstatic_geom < d.ngeom && gl[d.nstatic_geom] == 0) d.nstatic_geom++;
    }
    d.npair_pad = (d.npair + 7) & ~7;
    if (d.npair_pad == 0) d.npair_pad = 8;
    {   // derived tables: per-pair record and dof -> actuator map
        const int *g1 = m->i32("pair_geom1"), *g2 = m->i32("pair_geom2"), *cd = m->i32("pair_condim"), *gl = m->i32("geom_link"), *ad = m->i32("act_dof");
        const double *fr = m->f64("pair_friction"), *sr = m->f64("pair_solref"), *si = m->f64("pair_solimp"), *iw = m->f64("geom_invweight");
        std::vector<float> rec((size_t)std::max(d.npair, 1) * 16, 0.f);
        for (int p = 0; p < d.npair; p++) {
            float *r = rec.data() + 16 * p;
            r[0] = (float)cd[p]; r[1] = (float)gl[g1[p]]; r[2] = (float)gl[g2[p]]; r[3] = (float)(iw[2 * g1[p]] + iw[2 * g2[p]]);
            for (int j = 0; j < 5; j++) r[4 + j] = (float)fr[5 * p + j];
            // [9], [10]: what the contact rows need of solref and solimp's dmax, formed here in double: B = 2 / (dmax timeconst), K = 1 / (dmax^2 timeconst^2 dampratio^2)
            const double dmax = std::min(std::max(si[5 * p + 1], (double)HSR_MINIMP), (double)HSR_MAXIMP), tc = sr[2 * p], dr = sr[2 * p + 1];
            r[9] = (float)(2.0 / (dmax * tc)); r[10] = (float)(1.0 / (dmax * dmax * tc * tc * dr * dr));
            for (int j = 0; j < 5; j++) r[11 + j] = (float)si[5 * p + j];
        }
        float *drec; if ((rc = dalloc(b, &drec, rec.size()))) return rc;
        HIPCHK(hipMemcpy(drec, rec.data(), rec.size() * sizeof(float), hipMemcpyHostToDevice));
        d.pair_rec = drec;
        std::vector<int> da(std::max(d.nv, 1), -1);
        for (int a = 0; a < d.nu; a++) da[ad[a]] = a;
        int *dda; if ((rc = dalloc(b, &dda, da.size()))) return rc;
        HIPCHK(hipMemcpy(dda, da.data(), da.size() * sizeof(int), hipMemcpyHostToDevice));
        d.dof_act = dda;
    }
    {   // packed collision constants: one 32-float record per geom, one 8-float record per candidate pair
        const int *g1 = m->i32("pair_geom1"), *g2 = m->i32("pair_geom2"), *fn = m->i32("pair_fn"), *sl = m->i32("pair_slot");
        const int *gl = m->i32("geom_link"), *gt = m->i32("geom_type"), *ma = m->i32("geom_meshadr"), *mn = m->i32("geom_meshnum");
        const double *gp = m->f64("geom_pos"), *gq = m->f64("geom_quat"), *gs = m->f64("geom_size"), *gb = m->f64("geom_aabb"), *gr = m->f64("geom_rbound");
        std::vector<float> rec((size_t)std::max(d.npair, 1) * 8, 0.f), grec((size_t)std::max(d.ngeom, 1) * 32, 0.f);
        for (int p = 0; p < d.npair; p++) {
            float *r = rec.data() + 8 * p;
            r[0] = (float)(g1[p] + 256 * (gt[g1[p]] == GEOM_PLANE ? 1 : 0)); r[1] = (float)g2[p]; r[2] = (float)gr[g1[p]]; r[3] = (float)gr[g2[p]];   // [0]: geom1 | plane flag << 8
            r[4] = (float)fn[p]; r[5] = (float)sl[p]; r[6] = (float)(sl[p + 1] - sl[p]); r[7] = (float)gt[g1[p]];
        }
        for (int gg = 0; gg < d.ngeom; gg++) {
            // seven float4: link type nvert meshadr | lpos rbound | lmat[0..3] | lmat[4..7] | lmat[8] size | aabb centre - | aabb half -
            float *o = grec.data() + 32 * gg, lm[9];
            quat2mat_h(gq + 4 * gg, lm);
            o[0] = (float)gl[gg]; o[1] = (float)gt[gg]; o[2] = (float)mn[gg]; o[3] = (float)ma[gg];
            for (int k = 0; k < 3; k++) o[4 + k] = (float)gp[3 * gg + k];
            o[7] = (float)gr[gg];
            for (int k = 0; k < 9; k++) o[8 + k] = lm[k];
            for (int k = 0; k < 3; k++) o[17 + k] = (float)gs[3 * gg + k];
            for (int k = 0; k < 3; k++) { o[20 + k] = (float)gb[6 * gg + k]; o[24 + k] = (float)gb[6 * gg + 3 + k]; }
        }
        float *dg; if ((rc = dalloc(b, &dg, rec.size()))) return rc;
        HIPCHK(hipMemcpy(dg, rec.data(), rec.size() * sizeof(float), hipMemcpyHostToDevice));
        d.pair_geo = dg;
        if ((rc = dalloc(b, &dg, grec.size()))) return rc;
        HIPCHK(hipMemcpy(dg, grec.data(), grec.size() * sizeof(float), hipMemcpyHostToDevice));
        d.geom_rec = dg;
    }
    if ((rc = upload_mats(b, &d.link_mat, m, "link_quat"))) return rc;
    if ((rc = upload_mats(b, &d.geom_mat, m, "geom_quat"))) return rc;
    if ((rc = upload_mats(b, &d.body_mat, m, "body_quat"))) return rc;
    d.any_damping = 0;
    { size_t cnt; const double *dmp = m->f64("dof_damping", &cnt); for (size_t i = 0; i < cnt; i++) if (dmp[i] > 0) d.any_damping = 1; }
    d.solimp_general = 0;
    for (const char *nm : {"dof_solimp", "pair_solimp"}) {
        size_t cnt; const double *si = m->f64(nm, &cnt);
        for (size_t i = 4; si && i < cnt; i += 5) { const double pw = si[i] < 1 ? 1 : si[i]; if (pw != 1 && pw != 2) d.solimp_general = 1; }
    }

    DevState &s = b->ds;
    const size_t N = (size_t)n_envs;
    s.N = n_envs;
    s.npair_sep = std::max(d.npair, 1);
#define DA(field, rows) if ((rc = dalloc(b, &s.field, (size_t)(rows) * N))) return rc;
    DA(qpos, d.nq) DA(qvel, d.nv) DA(ctrl, d.nu) DA(mocap, 3) DA(warm, d.nv) DA(time, 1)
    DA(done, 1) DA(bad, 1) DA(nsteps, 1)
    DA(xpos, 3 * d.nlink) DA(xmat, 9 * d.nlink) DA(lvel, 6 * d.nlink)
    s.kstride = (9 * d.nv + 15 * d.nlink + 15) & ~15;
    DA(kin_aos, s.kstride)
    DA(con, 8 * d.nslot) DA(ncon_pair, d.npair_pad) DA(sepax, 4 * std::max(d.npair, 1)) DA(septick, std::max(d.npair, 1)) DA(tick, 1) DA(pair_list, std::max(d.npair, 1))
    if ((rc = dalloc(b, &s.pair_count, (size_t)d.npair_pad))) return rc;
    if ((rc = dalloc(b, &s.pair_pack, (size_t)((d.npair_pad + 7) & ~7)))) return rc;
    if ((rc = dalloc(b, &s.geom_c, (size_t)8 * std::max(d.ngeom, 1)))) return rc;
    DA(M, d.nM) DA(qacc, d.nv) DA(qacc_smooth, d.nv) DA(qfrc_smooth, d.nv) DA(qfrc_constraint, d.nv)
    DA(ncon, 1) DA(nefc, 1) DA(niter, 1)
#undef DA
    if ((rc = dalloc(b, &s.phase_cyc, 32 + 40 * 8192))) return rc;
    if ((rc = dalloc(b, &s.capstat, 12))) return rc;
    if ((rc = dalloc(b, &s.trips, N))) return rc;
    if ((rc = dalloc(b, &b->d_slot_env, N + 64))) return rc;
    {   // work queue of the persistent kernel: up to QUEUE_ROUNDS rounds of one ticket per task (a task = the envs of one workgroup)
        const size_t tasks = (N + 1) / 2;
        if ((rc = dalloc(b, &s.q_head, QUEUE_ROUNDS))) return rc;
        if ((rc = dalloc(b, &s.q_wpos, QUEUE_ROUNDS))) return rc;
        if ((rc = dalloc(b, &s.q_items, (size_t)QUEUE_ROUNDS * tasks))) return rc;
        if ((rc = dalloc(b, &s.q_err, 1))) return rc;
        s.sq_cap = (int)N + 4096;
        if ((rc = dalloc(b, &s.sq_items, (size_t)s.sq_cap))) return rc;
        if ((rc = dalloc(b, &s.sq_ctl, 4))) return rc;
        s.solo_servers = 0; s.solo_trips_x4 = 14; s.solo_min_left = 40;
        s.q_chunk = 0;
        { const char *so = getenv("HSR_SOLO"); if (so) b->solo_servers = atoi(so); const char *st = getenv("HSR_SOLO_TRIPS"); if (st && atof(st) > 0) b->solo_trips = (float)atof(st); }
        const char *q = getenv("HSR_QUEUE"); if (q) b->queue = atoi(q) != 0;
        const char *mw = getenv("HSR_MPR_WARM"); if (mw) b->mpr_warm = atoi(mw) != 0;
        const char *qc = getenv("HSR_QUEUE_CHUNK"); if (qc && atoi(qc) > 0) { b->queue_chunk = atoi(qc); b->queue_chunk_set = true; }
    }
    s.slot_env = nullptr;
    { const char *sc = getenv("HSR_SCHEDULE"); b->schedule = !(sc && strcmp(sc, "0") == 0); }        // on unless HSR_SCHEDULE=0
    // cooperative solver geometry: 16 lanes per env when nv <= 16, else 32
    b->group = d.nv <= 16 ? 16 : 32;
    {
        const char *nbk = getenv("HSR_NARROW_BLOCKS");
        if (nbk && atoi(nbk) > 0) b->narrow_blocks = atoi(nbk);
        const char *ppw = getenv("HSR_PPW");
        if (ppw && atoi(ppw) > 0) b->pairs_per_wave = atoi(ppw);
    }
    if (d.nv > 32 || d.nq > 64 || d.nlink > NLMAX || d.nconmax > b->group || b->ds.kstride > 8 * 4 * b->group)
        return fail(HSR_EINVAL, "model exceeds the lane-group solver (nv <= 32, nlink <= 16, nconmax <= lanes per env)");
    {
        const int total = b->group == 16 ? MfLayout<16>(d.njmax, b->ds.kstride).total : MfLayout<32>(d.njmax, b->ds.kstride).total;
        b->mf_lds_bytes = (size_t)total * (64 / b->group) * sizeof(float);
        if (b->mf_lds_bytes > 160 * 1024) return fail(HSR_EINVAL, "model exceeds the LDS budget of the solver");
        if (b->mf_lds_bytes > 48 * 1024) {
            if (b->group == 16) HIPCHK(hipFuncSetAttribute((const void *)k_solve_mf<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->mf_lds_bytes));
            else HIPCHK(hipFuncSetAttribute((const void *)k_solve_mf<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->mf_lds_bytes));
        }
    }
    {
        d.kin3_match = kin3_matching_row(m);
        if (d.kin3_match) {      // kin3.h stages 12 floats per dof and the robot's block of M in the row-scalar region of an env (Kin3Stage): it has to fit
            const int *lf = m->i32("link_free"), *dl = m->i32("dof_link");
            int nrd = 0;
            for (int k = 0; k < d.nv; k++) nrd += lf[dl[k]] ? 0 : 1;
            const int stage = 12 * d.nv + ((nrd + 3) & ~3) * nrd;
            const int region = b->group == 16 ? PersistLayout<16>(d.njmax, b->ds.kstride, d.npair_pad, d.nlink, d.ngeom, d.nstatic_geom).oCnt : PersistLayout<32>(d.njmax, b->ds.kstride, d.npair_pad, d.nlink, d.ngeom, d.nstatic_geom).oCnt;
            if (stage > region) d.kin3_match = 0;
        }
        {   // (before an instance is chosen: the constant instances are matched on nfb too)
            // trailing free bodies: link l owns exactly the dofs [nv - 6 (k + 1), nv - 6 k), lin then ang
            const int *dn = m->i32("link_dofnum"), *lf = m->i32("link_free"), *da = m->i32("link_dofadr"), *dt = m->i32("dof_type"), *dl = m->i32("dof_link");
            int nfb = 0;
            for (int k = 0; 6 * (k + 1) <= d.nv; k++) {
                const int a0 = d.nv - 6 * (k + 1), l = dl[a0];
                bool fb = l > 0 && lf[l] && da[l] == a0 && dn[l] == 6;
                for (int j = 0; fb && j < 6; j++) fb = dl[a0 + j] == l && dt[a0 + j] == (j < 3 ? DOF_FREE_LIN : DOF_FREE_ANG);
                if (!fb) break;
                nfb++;
            }
            if (nfb * 28 * (64 / b->group) > 4 * 48) nfb = 0;                  // the per-body accumulators live in the box-box polygon scratch
            const char *nf = getenv("HSR_NFB");                                // diagnostic: HSR_NFB=0 keeps the per-contact assembly
            if (nf && atoi(nf) < nfb) nfb = atoi(nf) < 0 ? 0 : atoi(nf);
            d.nfb = nfb;
        }
        {   // hulls staged in LDS by the instances that know their tree at compile time (their kin2 table area is free: persist.h): the hulls of the
            // deepest links first (the fingers: what the hard envs run MPR on), smaller ones first within a link depth, while they fit
            const int *gl = m->i32("geom_link"), *gt = m->i32("geom_type"), *ma = m->i32("geom_meshadr"), *mn = m->i32("geom_meshnum"), *lp = m->i32("link_parent");
            std::vector<int> ldsv(std::max(d.ngeom, 1), -1), src;
            const char *nh = getenv("HSR_LDS_HULLS");
            const int crow = cfg_const_row(d);          // (the instance that will run: HSR_NO_CONST=1 and a model that matches no row take a generic one, whose kin2 table occupies the area)
            if (crow >= 0 && kKin3Checks[crow].i && !(nh && strcmp(nh, "0") == 0)) {
                const int budget = KIN2_FLOATS * d.nlink / 4;
                auto depth = [&](int l) { int k = 0; while (l > 0) { l = lp[l]; k++; } return k; };
                std::vector<int> order;
                for (int g = 0; g < d.ngeom; g++) if (gt[g] == GEOM_MESH && gl[g] > 0 && mn[g] > 0) order.push_back(g);
                std::stable_sort(order.begin(), order.end(), [&](int a, int bb) { const int da = depth(gl[a]), db = depth(gl[bb]); return da != db ? da > db : mn[a] < mn[bb]; });
                for (int g : order) if ((int)src.size() + mn[g] <= budget) { ldsv[g] = (int)src.size(); for (int k = 0; k < mn[g]; k++) src.push_back(ma[g] + k); }
            }
            int *dl_; if ((rc = dalloc(b, &dl_, ldsv.size()))) return rc;
            HIPCHK(hipMemcpy(dl_, ldsv.data(), ldsv.size() * sizeof(int), hipMemcpyHostToDevice));
            d.geom_ldsv = dl_;
            int *ds_; if ((rc = dalloc(b, &ds_, src.size() + 1))) return rc;
            if (!src.empty()) HIPCHK(hipMemcpy(ds_, src.data(), src.size() * sizeof(int), hipMemcpyHostToDevice));
            d.ldsv_src = ds_; d.nldsv = (int)src.size();
        }
        auto lds_total = [&](bool tg) { return (size_t)sizeof(float) * (b->group == 16 ? PersistLayout<16>(d.njmax, b->ds.kstride, d.npair_pad, d.nlink, d.ngeom, d.nstatic_geom, tg).total
                                                                                            : PersistLayout<32>(d.njmax, b->ds.kstride, d.npair_pad, d.nlink, d.ngeom, d.nstatic_geom, tg).total); };
        b->persist_lds_bytes = lds_total(false);
        {   // a model whose tables cost the eighth workgroup per CU (160 KB / 8 = 20480 B each, static LDS included) reads them from global memory
            hipFuncAttributes fa;
            persist_fn f0 = persist_kernel(d, b->group, false), f1 = persist_kernel(d, b->group, true);
            if (f1 && hipFuncGetAttributes(&fa, (const void *)f0) == hipSuccess && b->persist_lds_bytes + fa.sharedSizeBytes > 20480
                && hipFuncGetAttributes(&fa, (const void *)f1) == hipSuccess && lds_total(true) + fa.sharedSizeBytes <= 20480) {
                b->persist_tg = true;
                b->persist_lds_bytes = lds_total(true);
            }
            const char *tg = getenv("HSR_TABLES_GLOBAL");                    // diagnostic: 0 keeps the tables in LDS
            if (tg && strcmp(tg, "0") == 0 && b->persist_tg) { b->persist_tg = false; b->persist_lds_bytes = lds_total(false); }
        }
        // what the persistent kernel's lane maps and kinematics assume (kin2.h, persist.h); a model outside it runs the per-substep chain
        bool ok = d.nq <= b->group && d.nv <= b->group && d.nlink <= b->group && d.nlink <= NLMAX && d.ngeom <= 64 && d.npair < (1 << 14) && d.maxdepth <= 9;
        {   // the persistent kernel keeps every dof's chain to the root in one 64-bit register, 6 bits per dof (persist.h: anc_c)
            const int *dp = m->i32("dof_parent");
            for (int c = 0; ok && c < d.nv; c++) { int depth = 0; for (int k = c; k >= 0 && depth <= 10; k = dp[k]) depth++; if (depth > 10) ok = false; }
        }
        {
            const int *dn = m->i32("link_dofnum"), *lf = m->i32("link_free"), *lp = m->i32("link_parent"), *gl = m->i32("geom_link");
            for (int l = 1; l < d.nlink; l++) {
                if (!lf[l] && dn[l] > 3) ok = false;                          // at most three scalar joints per link record
                if (lf[l] && lp[l] != 0) ok = false;                          // free bodies hang off the world ...
                if (lf[lp[l]]) ok = false;                                    // ... and carry no children
            }
            for (int gi = d.nstatic_geom; gi < d.ngeom; gi++) if (gl[gi] == 0) ok = false;   // static geoms form a prefix of the geom list
        }
        if (b->persist_lds_bytes > 160 * 1024) ok = false;
        if (!persist_kernel(d, b->group, b->persist_tg)) ok = false;      // development builds carry one instance only
        b->persist_ok = ok;
        if (!ok) d.nfb = 0;
        const char *pe = getenv("HSR_PERSIST");
        b->persist = ok && !(pe && strcmp(pe, "0") == 0);
        if (ok && b->persist_lds_bytes > 48 * 1024)
            HIPCHK(hipFuncSetAttribute((const void *)persist_kernel(d, b->group, b->persist_tg), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->persist_lds_bytes));
    }
    b->solo_ok = b->persist_ok && persist_kernel_sv(d, b->group, b->persist_tg) != nullptr;          // an instance with the server path exists
    if (b->solo_ok && b->persist_lds_bytes > 48 * 1024)
        HIPCHK(hipFuncSetAttribute((const void *)persist_kernel_sv(d, b->group, b->persist_tg), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->persist_lds_bytes));
    if (b->persist_ok) {
        int pb = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pb, persist_kernel(d, b->group, b->persist_tg), 64, b->persist_lds_bytes) == hipSuccess && pb > 0
            && hipGetDeviceProperties(&prop, b->device) == hipSuccess) b->slots = pb * prop.multiProcessorCount;
    }
    if (getenv("HSR_DEBUG") && b->persist_ok) {
        int pb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&pb, persist_kernel(d, b->group, b->persist_tg), 64, b->persist_lds_bytes);
        hipFuncAttributes fb;
        if (hipFuncGetAttributes(&fb, (const void *)persist_kernel(d, b->group, b->persist_tg)) == hipSuccess)
            fprintf(stderr, "[hsrsim] k_env_step_mf<%d>: regs %d, static LDS %zu, dyn LDS %zu, scratch %zu -> %d workgroups per CU\n", b->group, fb.numRegs, fb.sharedSizeBytes, b->persist_lds_bytes, fb.localSizeBytes, pb);
    }
    {
        const size_t kb = (size_t)64 * (b->ds.kstride + 24 * d.nlink + 1) * sizeof(float);
        if (kb > 160 * 1024) return fail(HSR_EINVAL, "kinematics tile exceeds LDS");
        if (kb > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void *)k_kinematics, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kb));
    }
    b->stage_floats = N * (size_t)(std::max(std::max(std::max(d.nq + d.nv, 25), 7 * d.nslot), std::max(d.nv * d.nv, 9 * d.nlink)) + d.nu + d.nq + d.nv + 4) + 16;
    if ((rc = dalloc(b, &b->d_stage, b->stage_floats))) return rc;
    if ((rc = dalloc(b, &b->d_stage_u8, N))) return rc;
    if ((rc = dalloc(b, &b->d_stage_i32, N))) return rc;
    HIPCHK(hipEventCreate(&b->ev0));
    HIPCHK(hipEventCreate(&b->ev1));
    // initial state = mj_resetData
    float *d_q0;
    if ((rc = dalloc(b, &d_q0, (size_t)d.nq))) return rc;
    HIPCHK(hipMemcpy(d_q0, m->qpos0.data(), d.nq * sizeof(float), hipMemcpyHostToDevice));
    b->d_qpos0 = d_q0;
    hipLaunchKernelGGL(k_build_tables, grid1((size_t)std::max(d.ngeom, (d.npair_pad + 7) & ~7)), dim3(256), 0, b->stream, b->dm, b->ds);
    hipLaunchKernelGGL(k_reset, grid1(N), dim3(256), 0, b->stream, b->dm, b->ds, (const uint8_t *)nullptr, (const float *)nullptr, (const float *)d_q0, (const float *)nullptr, 0);
    HIPCHK(hipStreamSynchronize(b->stream));
    return HSR_OK;
}

extern "C" void hsr_batch_destroy(hsr_batch *b) {
    if (!b) return;
    hipSetDevice(b->device);
    if (b->stream) hipStreamSynchronize(b->stream);
    for (auto &kv : b->graphs) hipGraphExecDestroy(kv.second);
    for (void *p : b->allocs) hipFree(p);
    for (hipEvent_t ev : b->kev) hipEventDestroy(ev);
    for (auto &pr : b->klog) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    if (b->ev0) hipEventDestroy(b->ev0);
    if (b->ev1) hipEventDestroy(b->ev1);
    if (b->stream) hipStreamDestroy(b->stream);
    delete b;
}
#define NULLCHK(b) do { if (!(b)) return fail(HSR_EINVAL, "null batch"); } while (0)
extern "C" int hsr_batch_size(const hsr_batch *b) { NULLCHK(b); return b->N; }
extern "C" void *hsr_batch_stream(const hsr_batch *b) { return b ? (void *)b->stream : nullptr; }
static int queue_error(hsr_batch *b);
extern "C" int hsr_batch_sync(hsr_batch *b) { NULLCHK(b); HIPCHK(hipSetDevice(b->device)); HIPCHK(hipStreamSynchronize(b->stream)); return queue_error(b); }
extern "C" int hsr_batch_set_profiling(hsr_batch *b, int on) { NULLCHK(b); b->profiling = on == 1; b->kernel_log = on == 2; return HSR_OK; }
// durations (ms) of the persistent-kernel launches logged since the last call (hsr_batch_set_profiling(b, 2)); synchronises the stream
extern "C" int hsr_batch_kernel_times(hsr_batch *b, float *out_ms, int cap) {
    NULLCHK(b);
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipStreamSynchronize(b->stream));
    int n = 0;
    for (auto &pr : b->klog) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess && out_ms && n < cap) out_ms[n] = ms;
        n++;
        hipEventDestroy(pr.first); hipEventDestroy(pr.second);
    }
    b->klog.clear();
    const int qe = queue_error(b);
    return qe ? qe : n;
}
extern "C" int hsr_batch_set_graph(hsr_batch *b, int on) { NULLCHK(b); b->use_graph = on != 0; return HSR_OK; }
__global__ void k_clear_margins(DevState s) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)s.npair_sep * s.N) {
        float *mg = s.sepax + (4 * (i / s.N) + 3) * s.N + i % s.N;
        if (*mg < 0.f) { mg[-(ptrdiff_t)s.N] = 0.f; mg[-2 * (ptrdiff_t)s.N] = 0.f; mg[-3 * (ptrdiff_t)s.N] = 0.f; }      // portal vertex ids, not a direction
        *mg = 0.f;
    }
}
static void clear_margins(hsr_batch *b) {
    hipLaunchKernelGGL(k_clear_margins, grid1((size_t)b->ds.npair_sep * b->N), dim3(256), 0, b->stream, b->ds);
}
extern "C" int hsr_batch_set_persistent(hsr_batch *b, int on) {
    NULLCHK(b);
    const bool want = on != 0 && b->persist_ok;
    if (want && !b->persist) { hipSetDevice(b->device); clear_margins(b); }      // the per-substep chain does not maintain the margins
    b->persist = want;
    return b->persist ? 1 : 0;
}
extern "C" int hsr_batch_is_persistent(const hsr_batch *b) {
    NULLCHK(b);
    if (!b->persist) return 0;
    const int row = cfg_const_row(b->dm);
    return 1 | (row >= 0 ? 2 : 0) | ((row >= 0 && kKin3Checks[row].i) ? 4 : 0);
}
extern "C" int hsr_batch_set_debug(hsr_batch *b, int on) { if (!b) return fail(HSR_EINVAL, "null batch"); b->debug_store = (on & 1) != 0; b->test_hooks = on & (6 | 16 | 32 | 64 | 128); return HSR_OK; }
extern "C" int hsr_batch_set_schedule(hsr_batch *b, int on) { NULLCHK(b); b->schedule = on != 0; return HSR_OK; }
extern "C" int hsr_batch_set_mpr_warm(hsr_batch *b, int on) {
    NULLCHK(b);
    HIPCHK(hipSetDevice(b->device));
    if ((on != 0) != b->mpr_warm) clear_margins(b);
    b->mpr_warm = on != 0;
    return HSR_OK;
}
extern "C" int hsr_batch_set_solo(hsr_batch *b, int servers, float trips) {
    NULLCHK(b);
    if (servers < 0 || trips < 0.f) return fail(HSR_EINVAL, "hsr_batch_set_solo: servers >= 0, trips >= 0");
    b->solo_servers = servers;
    if (trips > 0.f) b->solo_trips = trips;
    return b->solo_ok ? HSR_OK : 1;          // 1: accepted, but this model's kernel instance has no server path (the setting has no effect)
}
extern "C" int hsr_batch_solo_handovers(hsr_batch *b, int *out) {
    if (!b || !out) return fail(HSR_EINVAL, "null argument");
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipMemcpy(out, b->ds.sq_ctl + 1, sizeof(int), hipMemcpyDeviceToHost));
    return queue_error(b);
}
extern "C" int hsr_batch_set_queue(hsr_batch *b, int mode, int chunk) {
    NULLCHK(b);
    if (mode < -1 || mode > 1 || chunk < 0) return fail(HSR_EINVAL, "hsr_batch_set_queue: mode -1 / 0 / 1, chunk >= 0");
    b->queue = mode;
    if (chunk > 0) { b->queue_chunk = chunk; b->queue_chunk_set = true; }
    return HSR_OK;
}
// the work queue's watchdog (persist.h: q_claim) tripped in some launch since the last check: the flag is sticky on the device (no launch
// clears it) and only this function resets it, after reading it - every synchronising entry point ends with it
static int queue_error(hsr_batch *b) {
    int err = 0;
    if (b->ds.q_err && hipMemcpy(&err, b->ds.q_err, sizeof err, hipMemcpyDeviceToHost) == hipSuccess && err) {
        hipMemset(b->ds.q_err, 0, sizeof err);
        return fail(HSR_EDEVICE, "persistent kernel: a work-queue ticket was never served (launch drained by its watchdog)");
    }
    return HSR_OK;
}
extern "C" int hsr_batch_set_goals(hsr_batch *b, int n, const int *body_a, const int *body_b, const float *dist) {
    if (!b || n < 0 || n > 4 || (n > 0 && (!body_a || !body_b || !dist))) return fail(HSR_EINVAL, "hsr_batch_set_goals: 0..4 terms");
    for (int k = 0; k < n; k++)
        if (body_a[k] < 0 || body_a[k] >= b->dm.nbody || body_b[k] < 0 || body_b[k] >= b->dm.nbody) return fail(HSR_EINVAL, "hsr_batch_set_goals: body id out of range");
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipStreamSynchronize(b->stream));
    b->ds.ngoal = n;
    for (int k = 0; k < n; k++) { b->ds.goal_a[k] = body_a[k]; b->ds.goal_b[k] = body_b[k]; b->ds.goal_d[k] = dist[k]; }
    for (auto &kv : b->graphs) hipGraphExecDestroy(kv.second);      // captured launches carry the old terms
    b->graphs.clear();
    return HSR_OK;
}
extern "C" int hsr_batch_cap_counts(hsr_batch *b, unsigned long long *out) {
    if (!b || !out) return fail(HSR_EINVAL, "null argument");
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipMemcpy(out, b->ds.capstat, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(b->ds.capstat, 0, 4 * sizeof(unsigned long long)));
    return queue_error(b);
}
extern "C" int hsr_batch_cap_histogram(hsr_batch *b, unsigned long long *out) {
    if (!b || !out) return fail(HSR_EINVAL, "null argument");
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipMemcpy(out, b->ds.capstat + 4, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(b->ds.capstat + 4, 0, 8 * sizeof(unsigned long long)));
    return HSR_OK;
}

// per-env Newton iterations over the last (up to) 100 substeps of the previous persistent launch: what k_schedule packs by
extern "C" int hsr_batch_newton_trips(hsr_batch *b, int32_t *out) {
    if (!b || !out) return fail(HSR_EINVAL, "null argument");
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipMemcpy(out, b->ds.trips, (size_t)b->N * sizeof(int32_t), hipMemcpyDeviceToHost));
    return HSR_OK;
}
// the packing the last persistent launch ran with: out[slot] = env of lane group `slot % (64 / group)` of task `slot / (64 / group)`, -1 = empty
extern "C" int hsr_batch_packing(hsr_batch *b, int32_t *out) {
    if (!b || !out) return fail(HSR_EINVAL, "null argument");
    if (!b->persist || !b->schedule || !b->d_slot_env) return fail(HSR_EINVAL, "hsr_batch_packing: no packed persistent launch on this batch");
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipStreamSynchronize(b->stream));
    const int epb = 64 / b->group, slots = (b->N + epb - 1) / epb * epb;
    HIPCHK(hipMemcpy(out, b->d_slot_env, sizeof(int32_t) * slots, hipMemcpyDeviceToHost));
    return queue_error(b);
}

// one substep of the per-substep chain = 4 launches on the batch stream (the persistent kernel needs none of them)
static void launch_substep(hsr_batch *b, int mode, int goal_body, float geofence, int debug, hipStream_t st, bool timed) {
    const int N = b->N;
    auto rec = [&](void) { if (timed) { hipEvent_t ev; hipEventCreate(&ev); hipEventRecord(ev, st); b->kev.push_back(ev); } };
    rec();
    hipLaunchKernelGGL(k_kinematics, dim3((N + 63) / 64), dim3(64), (size_t)64 * (b->ds.kstride + 24 * b->dm.nlink + 1) * sizeof(float), st, b->dm, b->ds);
    rec();
    if (b->dm.npair > 0) {
        hipLaunchKernelGGL(k_cull, dim3((N + 63) / 64, (b->dm.npair + b->pairs_per_wave - 1) / b->pairs_per_wave), dim3(64), 0, st, b->dm, b->ds);
        hipLaunchKernelGGL(k_narrow, dim3(b->narrow_blocks), dim3(64), 0, st, b->dm, b->ds);
    }
    rec();
    if (b->group == 16) hipLaunchKernelGGL(k_solve_mf<16>, dim3((N + 3) / 4), dim3(64), b->mf_lds_bytes, st, b->dm, b->ds, mode, goal_body, geofence, debug);
    else hipLaunchKernelGGL(k_solve_mf<32>, dim3((N + 1) / 2), dim3(64), b->mf_lds_bytes, st, b->dm, b->ds, mode, goal_body, geofence, debug);
    rec();
}

extern "C" int hsr_batch_forward(hsr_batch *b) {
    NULLCHK(b);
    HIPCHK(hipSetDevice(b->device));
    hipLaunchKernelGGL(k_clear_done, grid1(b->N), dim3(256), 0, b->stream, b->ds);
    launch_substep(b, 0, -1, 0.f, 1, b->stream, false);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(b->stream));
    return HSR_OK;
}

extern "C" int hsr_batch_reset(hsr_batch *b, const uint8_t *mask, const float *qpos0, const float *mocap) {
    NULLCHK(b);
    HIPCHK(hipSetDevice(b->device));
    const size_t N = b->N;
    const int nq = b->dm.nq;
    float *d_q = nullptr, *d_m = nullptr, *d_q0m = b->d_qpos0;
    size_t off = 0;
    if (qpos0) { d_q = b->d_stage + off; off += N * nq; HIPCHK(hipMemcpyAsync(d_q, qpos0, N * nq * sizeof(float), hipMemcpyHostToDevice, b->stream)); }
    if (mocap) { d_m = b->d_stage + off; off += N * 3; HIPCHK(hipMemcpyAsync(d_m, mocap, N * 3 * sizeof(float), hipMemcpyHostToDevice, b->stream)); }
    if (off > b->stage_floats) return fail(HSR_EINVAL, "staging overflow in reset");
    if (mask) HIPCHK(hipMemcpyAsync(b->d_stage_u8, mask, N, hipMemcpyHostToDevice, b->stream));
    hipLaunchKernelGGL(k_reset, grid1(N), dim3(256), 0, b->stream, b->dm, b->ds, mask ? (const uint8_t *)b->d_stage_u8 : (const uint8_t *)nullptr,
                       (const float *)d_q, (const float *)d_q0m, (const float *)d_m, mask ? 1 : 0);
    // the forward pass concerns the reset envs only (MujocoEnv.reset() touches one env, hsr/mujoco_env.py:83-85): with a mask the
    // others stay parked as "done" for that pass, as in hsr_batch_reset_dev
    launch_substep(b, 0, -1, 0.f, 1, b->stream, false);
    hipLaunchKernelGGL(k_clear_done, grid1(N), dim3(256), 0, b->stream, b->ds);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(b->stream));
    return HSR_OK;
}

// the forward pass after a masked reset concerns the reset envs only (the reference's reset() touches one env): the others are
// parked as "done" for that pass by k_reset, so its kernels skip them (whole waves return when none of their envs was reset)
// device-pointer reset: envs with d_mask[e] != 0 (or, when d_mask == NULL, the envs whose done flag was
// latched by the last step) restart from d_qpos0[e] / d_mocap[e]; asynchronous; followed by forward.
extern "C" int hsr_batch_reset_dev(hsr_batch *b, const uint8_t *d_mask, const float *d_qpos0, const float *d_mocap) { NULLCHK(b);
    HIPCHK(hipSetDevice(b->device));
    const size_t N = b->N;
    // k_reset with park = 1 does all three jobs in one launch: mask = done flags (d_mask == NULL), reset of the masked envs, and the
    // unmasked ones parked as "done" for the forward pass that follows
    hipLaunchKernelGGL(k_reset, grid1(N), dim3(256), 0, b->stream, b->dm, b->ds, d_mask, d_qpos0, (const float *)b->d_qpos0, d_mocap, 1);
    launch_substep(b, 0, -1, 0.f, 0, b->stream, false);
    hipLaunchKernelGGL(k_clear_done, grid1(N), dim3(256), 0, b->stream, b->ds);
    HIPCHK(hipGetLastError());
    return HSR_OK;
}

static int to_device_soa(hsr_batch *b, float *dst, const float *host, int rows) {
    const size_t n = (size_t)rows * b->N;
    if (n > b->stage_floats) return fail(HSR_EINVAL, "staging overflow");
    HIPCHK(hipMemcpyAsync(b->d_stage, host, n * sizeof(float), hipMemcpyHostToDevice, b->stream));
    hipLaunchKernelGGL(k_aos_to_soa, grid1(n), dim3(256), 0, b->stream, dst, (const float *)b->d_stage, rows, b->N);
    HIPCHK(hipStreamSynchronize(b->stream));
    return HSR_OK;
}
static int to_host_aos(hsr_batch *b, float *host, const float *src, int rows) {
    const size_t n = (size_t)rows * b->N;
    if (n > b->stage_floats) return fail(HSR_EINVAL, "staging overflow");
    hipLaunchKernelGGL(k_soa_to_aos, grid1(n), dim3(256), 0, b->stream, b->d_stage, src, rows, b->N, rows, 0);
    HIPCHK(hipMemcpyAsync(host, b->d_stage, n * sizeof(float), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return HSR_OK;
}

extern "C" int hsr_batch_get_state(hsr_batch *b, float *time, float *qpos, float *qvel) { NULLCHK(b);
    HIPCHK(hipSetDevice(b->device));
    int rc;
    if (time && (rc = to_host_aos(b, time, b->ds.time, 1))) return rc;
    if (qpos && (rc = to_host_aos(b, qpos, b->ds.qpos, b->dm.nq))) return rc;
    if (qvel && (rc = to_host_aos(b, qvel, b->ds.qvel, b->dm.nv))) return rc;
    return queue_error(b);
}
extern "C" int hsr_batch_set_state(hsr_batch *b, const float *time, const float *qpos, const float *qvel) { NULLCHK(b);
    HIPCHK(hipSetDevice(b->device));
    clear_margins(b);                     // positions jump: the separation margins of the convex pairs are void
    int rc;
    if (time && (rc = to_device_soa(b, b->ds.time, time, 1))) return rc;
    if (qpos && (rc = to_device_soa(b, b->ds.qpos, qpos, b->dm.nq))) return rc;
    if (qvel && (rc = to_device_soa(b, b->ds.qvel, qvel, b->dm.nv))) return rc;
    return hsr_batch_forward(b);
}
extern "C" int hsr_batch_set_mocap(hsr_batch *b, const float *mocap) { NULLCHK(b); HIPCHK(hipSetDevice(b->device)); return to_device_soa(b, b->ds.mocap, mocap, 3); }
extern "C" int hsr_batch_set_warmstart(hsr_batch *b, const float *w) { NULLCHK(b); HIPCHK(hipSetDevice(b->device)); return to_device_soa(b, b->ds.warm, w, b->dm.nv); }
extern "C" int hsr_batch_get_warmstart(hsr_batch *b, float *w) { NULLCHK(b); HIPCHK(hipSetDevice(b->device)); return to_host_aos(b, w, b->ds.warm, b->dm.nv); }

// Wave packing of the persistent kernel (on by default; HSR_SCHEDULE=0 or hsr_batch_set_schedule(b, 0) keeps the identity packing).
// A launch ends with the wave that holds the hardest env (the one that needs the most Newton iterations per substep), and a wave
// advances at the pace of its hardest env while the others idle: so every one of the hardest envs gets a wave of its own, filled up
// with the easiest envs (which leave the Newton loop after one iteration), hardest waves dispatched first.  Hardness = the
// iterations an env ran in the last 100 substeps of its previous launch (DevState::trips).  Measured (r2, 8192 envs, the bench's
// freshly sampled ctrl per env-step - the worst case for a predictor: corr 0.3 from one env-step to the next,
// tools/exp_predict.py): cfg3 +0.5..1 % (one round of 2048 workgroups: only the packing counts), cfg4 +6 % (4096 workgroups over
// 1792 slots: the dispatch order counts too); with the packing computed from the state the env-step starts from it would be 15 %,
// and a policy whose actions are correlated from one env-step to the next comes closer to that.  Splitting the env-step into
// re-packed launches costs more than it gains (every launch then waits for its own slowest wave: +10 %).
// One workgroup sorts up to 8192 envs (bitonic, keys in LDS); larger batches are packed chunk by chunk.
// Results do not depend on the packing: no value of an env is ever combined with another env's.
enum { SCHED_CHUNK = 8192 };
// round 0 of the work queue holds every task in packing order (hard ones first); the other rounds are empty
__global__ void k_queue_init(DevState s, int T, int R) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R) { s.q_head[i] = 0; s.q_wpos[i] = i == 0 ? T : 0; }
    if (i < 4) s.sq_ctl[i] = i == 2 ? s.solo_servers : 0;          // tickets taken, items reserved, free servers, finished tasks
    if (s.solo_servers > 0) for (int k = i; k < s.sq_cap; k += gridDim.x * blockDim.x) s.sq_items[k] = -1;
    // q_err is NOT cleared here: a trip stays on record until the host has read it (queue_error), however many launches were enqueued since
    if (i < R * T) s.q_items[i] = i < T ? i : -1;
}
// The bitonic network with eight consecutive keys per thread in registers: exchanges at distance 1, 2, 4 stay inside the thread, 8 .. 256 inside
// the wave (one shuffle per key), and only 512 .. 4096 cross waves through LDS - 10 of the 91 stages need a barrier (round 3: all 91, 124 us of
// every env-step; the packing it computes saves 280 us of the cfg3 launch)
template <int J> __device__ __forceinline__ void sched_local_stage(unsigned (&v)[8], int tid, int k) {
    unsigned w[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int i = 8 * tid + r;
        const unsigned a = v[r], c = v[r ^ J];
        w[r] = (((i & J) == 0) == ((i & k) == 0)) ? (a > c ? a : c) : (a < c ? a : c);     // the lower index of a descending pair keeps the larger key
    }
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = w[r];
}
__global__ void __launch_bounds__(1024) k_schedule(DevState s, int epb, int *slot_env) {
    __shared__ unsigned key[SCHED_CHUNK];
    const int tid = threadIdx.x;
    const int e0 = blockIdx.x * SCHED_CHUNK, n = min(SCHED_CHUNK, s.N - e0);
    unsigned v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int i = 8 * tid + r;
        const int t = i < n ? min(s.trips[e0 + i], 0x1fffe) : 0;
        v[r] = i < n ? ((unsigned)(t + 1) << 13) | (unsigned)(SCHED_CHUNK - 1 - i) : 0u;       // descending: more iterations first, then lower index
    }
    for (int k = 2; k <= SCHED_CHUNK; k <<= 1) {
        for (int j = k >> 1; j >= 512; j >>= 1) {              // partner in another wave: keys laid out [register][thread], no bank conflicts
#pragma unroll
            for (int r = 0; r < 8; r++) key[1024 * r + tid] = v[r];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int i = 8 * tid + r;
                const unsigned a = v[r], c = key[1024 * r + (tid ^ (j >> 3))];
                v[r] = (((i & j) == 0) == ((i & k) == 0)) ? (a > c ? a : c) : (a < c ? a : c);
            }
            __syncthreads();
        }
        for (int j = (k >> 1) < 256 ? (k >> 1) : 256; j >= 8; j >>= 1) {      // partner in the same wave
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int i = 8 * tid + r;
                const unsigned a = v[r], c = (unsigned)__shfl_xor((int)v[r], j >> 3, 64);
                v[r] = (((i & j) == 0) == ((i & k) == 0)) ? (a > c ? a : c) : (a < c ? a : c);
            }
        }
        if (k >= 8) sched_local_stage<4>(v, tid, k);
        if (k >= 4) sched_local_stage<2>(v, tid, k);
        sched_local_stage<1>(v, tid, k);
    }
#pragma unroll
    for (int r = 0; r < 8; r++) key[8 * tid + r] = v[r];
    __syncthreads();
    const int nw = (n + epb - 1) / epb;
    for (int sl = threadIdx.x; sl < nw * epb; sl += blockDim.x) {
        const int w = sl / epb, j = sl % epb;
        int idx;                                                   // position in the sorted list
        if (j == 0) idx = w;
        else { const int r = (j - 1) * nw + w; idx = r < n - nw ? n - 1 - r : -1; }
        slot_env[(size_t)e0 / epb * epb + sl] = (idx >= 0 && idx < n) ? e0 + (SCHED_CHUNK - 1 - (int)(key[idx] & (SCHED_CHUNK - 1))) : -1;
    }
}

extern "C" int hsr_batch_step_dev(hsr_batch *b, const float *d_ctrl, int n_substeps, int goal_body, float geofence,
                                  float *d_obs, float *d_reward, uint8_t *d_done, int32_t *d_nsteps) {
    if (!b || !d_ctrl || n_substeps < 0) return fail(HSR_EINVAL, "bad arguments to hsr_batch_step");
    if (goal_body >= b->dm.nbody) return fail(HSR_EINVAL, "goal body out of range");
    HIPCHK(hipSetDevice(b->device));
    const int N = b->N;
    hipStream_t st = b->stream;
    if (b->profiling) {
        for (hipEvent_t ev : b->kev) hipEventDestroy(ev);
        b->kev.clear();
        HIPCHK(hipEventRecord(b->ev0, st));
    }
    const bool fused = b->persist && n_substeps > 0;      // the persistent kernel reads ctrl and writes obs / reward / done / nsteps itself
    if (!fused) hipLaunchKernelGGL(k_begin_step, grid1(N), dim3(256), 0, st, b->ds, d_ctrl, b->dm.nu);
    if (fused) {
        const int epb = 64 / b->group;
        if (b->profiling) { hipEvent_t ev; for (int k = 0; k < 3; k++) { hipEventCreate(&ev); hipEventRecord(ev, st); b->kev.push_back(ev); } }
        if (!b->d_dm) {
            int rc2 = dalloc(b, &b->d_dm, 1);
            if (rc2) return rc2;
            HIPCHK(hipMemcpy(b->d_dm, &b->dm, sizeof(DevModel), hipMemcpyHostToDevice));
        }
        const bool sched = b->schedule;
        if (sched) hipLaunchKernelGGL(k_schedule, dim3((N + SCHED_CHUNK - 1) / SCHED_CHUNK), dim3(1024), 0, st, b->ds, epb, b->d_slot_env);
        DevState dsl = b->ds;
        dsl.slot_env = sched ? b->d_slot_env : nullptr;
        const StepIO io{d_ctrl, d_obs, d_reward, d_done, d_nsteps};
        // more tasks than the GPU holds workgroups at once: persistent workgroups + the work queue (persist.h), else one task per workgroup
        const int T = (N + epb - 1) / epb;
        int chunk = b->queue_chunk;
        // many tasks per resident workgroup (65536 envs: eight) balance themselves: longer rounds there, fewer hand-overs through the state arrays
        // and fewer rebuilds of the item lists (measured at 65536 envs, cfg3: rounds of 20 / 50 / 100 / 300 substeps: 833 / 856 / 841 / 779 k env-steps/s)
        if (!b->queue_chunk_set && b->slots > 0 && T >= 4 * b->slots) chunk = 50;
        while ((n_substeps + chunk - 1) / chunk > QUEUE_ROUNDS) chunk *= 2;
        // solo servers need the queue (a hard env leaves its task at the end of a round) and the env -> slot table
        // (a hand-over ticket packs env | substep << 20 into one int that must stay non-negative: fewer than 2048 substeps, at most 2^20 envs - beyond
        // that the launch simply runs without servers)
        const bool solo = b->solo_ok && b->solo_servers > 0 && b->solo_servers <= 4096 && sched && b->slots > 0 && n_substeps >= 3 * chunk && n_substeps < 2048 && b->N <= (1 << 20)
                          && (T + b->solo_servers <= b->slots || 4 * b->solo_servers <= b->slots);
        const bool qon = b->slots > 0 && n_substeps >= 2 * chunk && (solo || b->queue == 1 || (b->queue < 0 && T > b->slots));
        int grid = T;
        dsl.solo_servers = 0;
        if (qon) {
            const int R = (n_substeps + chunk - 1) / chunk;
            dsl.q_chunk = chunk;
            grid = T < b->slots ? T : b->slots;
            if (solo) { dsl.solo_servers = b->solo_servers; dsl.solo_trips_x4 = (int)(4.f * b->solo_trips + 0.5f); dsl.solo_min_left = 2 * chunk; grid = std::min(b->slots, T + b->solo_servers); }
            hipLaunchKernelGGL(k_queue_init, grid1((size_t)R * T), dim3(256), 0, st, dsl, T, R);
        }
        hipEvent_t k0 = nullptr, k1 = nullptr;
        if (b->kernel_log) { hipEventCreate(&k0); hipEventCreate(&k1); hipEventRecord(k0, st); }
        hipLaunchKernelGGL(dsl.solo_servers > 0 ? persist_kernel_sv(b->dm, b->group, b->persist_tg) : persist_kernel(b->dm, b->group, b->persist_tg), dim3(grid), dim3(64), b->persist_lds_bytes, st, (const DevModel *)b->d_dm, dsl, n_substeps, goal_body, geofence, (b->debug_store ? 1 : 0) | (b->test_hooks & ~32) | (b->mpr_warm ? 0 : 8), io);
        if ((b->test_hooks & 32) && b->ds.q_err) HIPCHK(hipMemsetD32Async((hipDeviceptr_t)b->ds.q_err, 1, 1, st));      // tests: what q_claim's watchdog does when a ticket is never served
        if (b->kernel_log) { hipEventRecord(k1, st); b->klog.push_back({k0, k1}); }
        if (b->profiling) { hipEvent_t ev; hipEventCreate(&ev); hipEventRecord(ev, st); b->kev.push_back(ev); }   // slots 0,1 empty; slot 2 = the persistent kernel
    } else if (b->use_graph && !b->profiling && n_substeps > 0) {
        GraphKey key{n_substeps, goal_body, geofence};
        auto it = b->graphs.find(key);
        if (it == b->graphs.end()) {
            hipGraph_t graph;
            HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < n_substeps; i++) launch_substep(b, 1, goal_body, geofence, 0, st, false);
            HIPCHK(hipStreamEndCapture(st, &graph));
            hipGraphExec_t exec;
            HIPCHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            hipGraphDestroy(graph);
            if (b->graphs.size() >= 8) { for (auto &kv : b->graphs) hipGraphExecDestroy(kv.second); b->graphs.clear(); }
            it = b->graphs.emplace(key, exec).first;
        }
        HIPCHK(hipGraphLaunch(it->second, st));
    } else {
        for (int i = 0; i < n_substeps; i++) launch_substep(b, 1, goal_body, geofence, 0, st, b->profiling);
    }
    if (!fused) {
        if (d_obs) {
            const int nq = b->dm.nq, nv = b->dm.nv;
            hipLaunchKernelGGL(k_soa_to_aos, grid1((size_t)nq * N), dim3(256), 0, st, d_obs, (const float *)b->ds.qpos, nq, N, nq + nv, 0);
            hipLaunchKernelGGL(k_soa_to_aos, grid1((size_t)nv * N), dim3(256), 0, st, d_obs, (const float *)b->ds.qvel, nv, N, nq + nv, nq);
        }
        hipLaunchKernelGGL(k_end_step, grid1(N), dim3(256), 0, st, b->ds, d_reward, d_done, d_nsteps);
    }
    HIPCHK(hipGetLastError());
    if (b->profiling) {
        HIPCHK(hipEventRecord(b->ev1, st));
        HIPCHK(hipEventSynchronize(b->ev1));
        HIPCHK(hipEventElapsedTime(&b->last_total_ms, b->ev0, b->ev1));
        for (int k = 0; k < 3; k++) { b->last_kernel_ms[k] = 0; b->last_launches[k] = 0; }
        for (size_t i = 0; i + 3 < b->kev.size(); i += 4)
            for (int k = 0; k < 3; k++) { float ms = 0; hipEventElapsedTime(&ms, b->kev[i + k], b->kev[i + k + 1]); b->last_kernel_ms[k] += ms; b->last_launches[k]++; }
    }
    return HSR_OK;
}

extern "C" int hsr_batch_step(hsr_batch *b, const float *ctrl, int n_substeps, int goal_body, float geofence,
                              float *obs, float *reward, uint8_t *done, int32_t *nsteps) {
    if (!b || !ctrl) return fail(HSR_EINVAL, "bad arguments to hsr_batch_step");
    HIPCHK(hipSetDevice(b->device));
    const size_t N = b->N;
    const int nu = b->dm.nu, no = b->dm.nq + b->dm.nv;
    // staging layout: [ctrl N*nu | obs N*no | reward N]
    if (N * (size_t)(nu + no + 1) > b->stage_floats) return fail(HSR_EINVAL, "staging overflow in step");
    float *d_ctrl = b->d_stage, *d_obs = b->d_stage + N * nu, *d_rew = d_obs + N * no;
    HIPCHK(hipMemcpyAsync(d_ctrl, ctrl, N * nu * sizeof(float), hipMemcpyHostToDevice, b->stream));
    int rc = hsr_batch_step_dev(b, d_ctrl, n_substeps, goal_body, geofence, obs ? d_obs : nullptr, reward ? d_rew : nullptr,
                                done ? b->d_stage_u8 : nullptr, nsteps ? b->d_stage_i32 : nullptr);
    if (rc) return rc;
    if (obs) HIPCHK(hipMemcpyAsync(obs, d_obs, N * no * sizeof(float), hipMemcpyDeviceToHost, b->stream));
    if (reward) HIPCHK(hipMemcpyAsync(reward, d_rew, N * sizeof(float), hipMemcpyDeviceToHost, b->stream));
    if (done) HIPCHK(hipMemcpyAsync(done, b->d_stage_u8, N, hipMemcpyDeviceToHost, b->stream));
    if (nsteps) HIPCHK(hipMemcpyAsync(nsteps, b->d_stage_i32, N * sizeof(int32_t), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return queue_error(b);
}

extern "C" int hsr_batch_body_xpos(hsr_batch *b, int body_id, float *out) { NULLCHK(b);
    if (body_id < 0 || body_id >= b->dm.nbody) return fail(HSR_EINVAL, "body id out of range");
    HIPCHK(hipSetDevice(b->device));
    hipLaunchKernelGGL(k_body_xpos, grid1(b->N), dim3(256), 0, b->stream, b->dm, b->ds, body_id, b->d_stage);
    HIPCHK(hipMemcpyAsync(out, b->d_stage, (size_t)b->N * 3 * sizeof(float), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return HSR_OK;
}

static int obs_openai_launch(hsr_batch *b, const int *ids, float *d_out) {
    const DevModel &d = b->dm;
    for (int k = 0; k < 3; k++) if (ids[k] < 0 || ids[k] >= d.nbody || b->model->i32("body_mocap")[ids[k]]) return fail(HSR_EINVAL, "obs_openai: bad body id");
    if (ids[3] < 0 || ids[3] >= d.nq || ids[4] < 0 || ids[4] >= d.nq || ids[5] < 0 || ids[5] >= d.nv || ids[6] < 0 || ids[6] >= d.nv)
        return fail(HSR_EINVAL, "obs_openai: bad joint address");
    HIPCHK(hipSetDevice(b->device));
    hipLaunchKernelGGL(k_obs_openai, grid1(b->N), dim3(256), 0, b->stream, b->dm, b->ds, ids[0], ids[1], ids[2], ids[3], ids[4], ids[5], ids[6], d.timestep, d_out);
    HIPCHK(hipGetLastError());
    return HSR_OK;
}
extern "C" int hsr_batch_obs_openai_dev(hsr_batch *b, const int *ids, float *d_out) { return obs_openai_launch(b, ids, d_out); }
extern "C" int hsr_batch_obs_openai(hsr_batch *b, const int *ids, float *out) {
    int rc = obs_openai_launch(b, ids, b->d_stage);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(out, b->d_stage, (size_t)b->N * 25 * sizeof(float), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return HSR_OK;
}

extern "C" int hsr_batch_bad_state(hsr_batch *b, uint8_t *out) { NULLCHK(b);
    HIPCHK(hipSetDevice(b->device));
    std::vector<int> tmp(b->N);
    HIPCHK(hipMemcpyAsync(tmp.data(), b->ds.bad, (size_t)b->N * sizeof(int), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    int any = 0;
    for (int i = 0; i < b->N; i++) { out[i] = (uint8_t)(tmp[i] != 0); any |= tmp[i]; }
    const int qe = queue_error(b);        // a drained launch outranks a diverged env: its envs stopped mid env-step
    return qe ? qe : (any ? HSR_EBADSTATE : HSR_OK);
}

extern "C" int hsr_batch_get_field(hsr_batch *b, int field, float *out) { NULLCHK(b);
    HIPCHK(hipSetDevice(b->device));
    const DevModel &d = b->dm;
    const size_t N = b->N;
    switch (field) {
    case HSR_F_XPOS: return to_host_aos(b, out, b->ds.xpos, 3 * d.nlink);
    case HSR_F_XMAT: return to_host_aos(b, out, b->ds.xmat, 9 * d.nlink);
    case HSR_F_QACC: return to_host_aos(b, out, b->ds.qacc, d.nv);
    case HSR_F_QACC_SMOOTH: return to_host_aos(b, out, b->ds.qacc_smooth, d.nv);
    case HSR_F_QFRC_SMOOTH: return to_host_aos(b, out, b->ds.qfrc_smooth, d.nv);
    case HSR_F_QFRC_CONSTRAINT: return to_host_aos(b, out, b->ds.qfrc_constraint, d.nv);
    case HSR_F_M:
        hipLaunchKernelGGL(k_expand_M, grid1(N), dim3(256), 0, b->stream, b->ds, b->d_stage, d.nv);
        HIPCHK(hipMemcpyAsync(out, b->d_stage, N * d.nv * d.nv * sizeof(float), hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
        return HSR_OK;
    case HSR_F_NCON: case HSR_F_NEFC: case HSR_F_NITER: {
        const int *src = field == HSR_F_NCON ? b->ds.ncon : (field == HSR_F_NEFC ? b->ds.nefc : b->ds.niter);
        hipLaunchKernelGGL(k_i32_to_f32, grid1(N), dim3(256), 0, b->stream, b->d_stage, src, N);
        HIPCHK(hipMemcpyAsync(out, b->d_stage, N * sizeof(float), hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
        return HSR_OK; }
    case HSR_F_CONTACT:
        hipLaunchKernelGGL(k_contacts_out, grid1(N), dim3(256), 0, b->stream, b->dm, b->ds, b->d_stage);
        HIPCHK(hipMemcpyAsync(out, b->d_stage, N * d.nslot * 7 * sizeof(float), hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
        return HSR_OK;
    default: return fail(HSR_EINVAL, "unknown field");
    }
}

// diagnostic builds (-DHSR_PHASE_TIMING): read and clear the per-phase cycle sums of k_solve_g
extern "C" int hsr_batch_phase_cycles(hsr_batch *b, unsigned long long *out /*[32]*/) { NULLCHK(b);
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipMemcpy(out, b->ds.phase_cyc, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(b->ds.phase_cyc, 0, 32 * sizeof(unsigned long long)));
    return HSR_OK;
}

// diagnostic builds: per-workgroup (start, end) s_memrealtime stamps and HW_ID / XCC_ID of the last persistent launch
extern "C" int hsr_batch_block_times(hsr_batch *b, unsigned long long *out, int nblocks) { NULLCHK(b);
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipStreamSynchronize(b->stream));
    if (nblocks > 8192) nblocks = 8192;
    if (nblocks < 0) {          // lifetime build: the per-env stamps (solve_g.h ENV_STAMP), 2 x 8192 values
        HIPCHK(hipMemcpy(out, b->ds.phase_cyc + 32 + 40 * 4096, (size_t)2 * 8192 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        HIPCHK(hipMemset(b->ds.phase_cyc + 32 + 40 * 4096, 0, (size_t)2 * 8192 * sizeof(unsigned long long)));
        return HSR_OK;
    }
    HIPCHK(hipMemcpy(out, b->ds.phase_cyc + 32, (size_t)nblocks * 40 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return HSR_OK;
}

extern "C" int hsr_batch_last_timing(hsr_batch *b, float *total_ms, float *kernel_ms, int *launches) { NULLCHK(b);
    if (total_ms) *total_ms = b->last_total_ms;
    for (int k = 0; k < 3; k++) { if (kernel_ms) kernel_ms[k] = b->last_kernel_ms[k]; if (launches) launches[k] = b->last_launches[k]; }
    return HSR_OK;
}
