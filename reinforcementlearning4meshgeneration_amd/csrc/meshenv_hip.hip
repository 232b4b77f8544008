// meshenv_hip.hip -- C-ABI (include/meshenv.h) over the wave-per-environment kernels.
//
// Host side: owns the HBM state of a batch of environments, launches k_init_domains / k_reset / k_step on
// the handle's stream.  Nothing here computes environment results on the host: without a working GPU every
// entry point fails with MESHENV_E_HIP.
#include "../../include/meshenv.h"

#include <hip/hip_runtime.h>
#include <link.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "meshenv_actor.h"
#include "meshenv_domgen.h"
#include "meshenv_kernels.h"
#include "meshenv_quality.h"
#include "meshenv_smooth.h"
#include "meshenv_samples.h"
#include "meshenv_fused.h"

using namespace meshenv;

namespace {
thread_local std::string g_create_error;

// ---- glibc's atan2, restated (csrc/meshenv_libm.h): the 241 x 7 table comes out of the libm image of this process.
// A record is {x_i, atan(x_i), 1 / (1 + x_i^2), ...} with x_i within 1/256 of (i + 16) / 256; the scan accepts the first
// 8-byte-aligned run of 241 such records in a readable segment of libm.so.
struct AtanHost {
    std::vector<double> cij;   // empty: not found
    int mode = 0;              // 2 restated glibc validated against this libm, 1 correctly rounded (atan2_cr), 0 ocml only
};

static bool atan_row_ok(const double *r, int i)
{
    const double x = r[0];
    if (!(std::fabs(x - (i + 16) / 256.0) < 1.0 / 256.0)) return false;
    return std::fabs(r[1] - std::atan(x)) < 1e-15 && std::fabs(r[2] * (1.0 + x * x) - 1.0) < 1e-12;
}

static int atan_scan_cb(struct dl_phdr_info *info, size_t, void *data)
{
    AtanHost *st = (AtanHost *)data;
    if (!st->cij.empty() || !info->dlpi_name || !std::strstr(info->dlpi_name, "libm.so")) return 0;
    const size_t bytes = sizeof(double) * kAtanRows * kAtanCols;
    for (int k = 0; k < info->dlpi_phnum; k++) {
        const ElfW(Phdr) *ph = &info->dlpi_phdr[k];
        if (ph->p_type != PT_LOAD || !(ph->p_flags & PF_R) || ph->p_memsz < bytes) continue;
        const unsigned char *base = (const unsigned char *)(info->dlpi_addr + ph->p_vaddr);
        size_t o = (8 - ((uintptr_t)base & 7)) & 7;
        for (; o + bytes <= ph->p_memsz; o += 8) {
            double r[3];
            std::memcpy(r, base + o, sizeof r);
            if (!(r[0] > 0.06 && r[0] < 0.07) || !atan_row_ok(r, 0)) continue;
            std::vector<double> t((size_t)kAtanRows * kAtanCols);
            std::memcpy(t.data(), base + o, bytes);
            bool all = true;
            for (int i = 0; i < kAtanRows && all; i++) all = atan_row_ok(t.data() + (size_t)kAtanCols * i, i);
            if (all) { st->cij.swap(t); return 1; }
        }
    }
    return 0;
}

// atan2_glibc against the atan2 of this process on 2^18 arguments: all four octants, both the polynomial (u < 1/16) and
// the table branch, lattice (4-decimal) and unrestricted coordinates, and the half-quantum angles the front smoother builds.
static bool atan_validate(const std::vector<double> &cij)
{
    unsigned long long s = 0x9E3779B97F4A7C15ULL;
    auto next = [&s]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    auto unit = [&next]() { return (double)(next() >> 11) / 9007199254740992.0; };
    for (int n = 0; n < (1 << 18); n++) {
        double y = 6.0 * unit() - 3.0, x = 6.0 * unit() - 3.0;
        const int kind = n & 3;
        if (kind == 1) { y = std::round(y * 1e4) / 1e4; x = std::round(x * 1e4) / 1e4; }
        else if (kind == 2) y = std::ldexp(y, (int)(next() % 60) - 30);
        else if (kind == 3) {
            const double h = ((double)(next() % 62832) + 0.5) * 1e-4, sc = 0.01 + 3.0 * unit();
            y = -std::sin(h) * sc; x = std::cos(h) * sc;
        }
        bool ok = false;
        const double mine = atan2_glibc(y, x, cij.data(), ok);
        if (ok && f64_bits(mine) != f64_bits(std::atan2(y, x))) return false;
    }
    return true;
}

static const AtanHost &atan_host()
{
    static AtanHost st;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *off = std::getenv("MESHENV_LIBM_EXACT");   // "0": ocml's atan2 everywhere (as before round 3)
        if (off && off[0] == '0') { st.mode = 0; return; }
        volatile double one = 1.0;
        (void)std::atan2(one, one);   // keeps libm's atan2 linked and resolved
        dl_iterate_phdr(atan_scan_cb, &st);
        st.mode = (!st.cij.empty() && atan_validate(st.cij)) ? 2 : 1;
    });
    return st;
}

// table + mode into the current device's copy of the module globals (13.5 KB; every handle creation and selftest does it)
static hipError_t upload_atan_state()
{
    const AtanHost &st = atan_host();
    hipError_t e = hipSuccess;
    if (st.mode == 2) e = hipMemcpyToSymbol(HIP_SYMBOL(g_atan_cij), st.cij.data(), sizeof(double) * st.cij.size());
    const int mode = st.mode;
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_atan_mode), &mode, sizeof mode);
    return e;
}
}

struct MeshEnv {
    int device = 0;
    hipStream_t stream = nullptr;
    DevState S{};
    int n_envs = 0, n_domains = 0, cap = 0, max_ring = 0;
    uint64_t steps_done = 0;  // vector steps executed so far (index of the next step)
    DevCold cold{};  // host copy of *S.cold
    bool default_params = true;  // geometry constants are the reference's: literal-constant kernel instantiations
    size_t lds = 0;
    int n_cu = 256;       // compute units of the device (hipDeviceAttributeMultiprocessorCount)
    bool spec = false;    // single-step kernel = k_step_spec (speculative extraction, no workgroup barrier)
    int group = 1;        // environments (wavefronts) per workgroup of the single-step kernel
    size_t group_lds = 0;
    int2 *env_lds = nullptr;   // ragged LDS packing of the CU-group kernel: [n_envs] (byte offset in the workgroup, ring slots), or null
    int ho_off = 0;            // ragged: offset of the hand-over blocks
    std::vector<int2> env_lds_host;
    std::vector<int32_t> dom_off_host, env_dom_host;
    std::vector<void *> allocs;
    std::string err;
    int timing = 0;              // 0 = off, k = bracket every other group of k consecutive launches
    long long launch_count = 0;
    std::vector<hipEvent_t> ev;  // 2 * MESHENV_TIMING_POOL events, created on first use
    int32_t *smooth_sweeps = nullptr;  // [E], allocated by the first meshenv_smooth: what the smoother did per env
    int32_t *front_code = nullptr;     // [E] outcome of the front smoother (gates the interior pass of the same call)
    double *front_tab = nullptr;       // [kFtTotal] the front smoother's tan / cos values from the host's libm
    bool smooth_ready = false, move_ready = false;   // set only after every allocation / attribute call succeeded
    int libm_exact = -1;               // pow2_glibc == the running libm's pow(x, 2.0) on the validation set (-1: not checked yet)
    Reselect *pend = nullptr;          // [E] selection parked by the candidate rebuild (csrc/meshenv_smooth.h)
    float *pend_obs = nullptr;         // [E][18]
    bool reselect_pending = false;     // a rebuild ran since the last step kernel
    bool front_moved = false;          // a front smoother ran since the last full reset: rings may hold off-lattice vertices
    bool smooth_final_ready = false;
    bool fused_ready = false;          // k_step_group_actor's LDS attribute set
    bool fused_T_ready = false;        // k_step_group_actor_T's (-DMESHENV_DEV build only)
    int stage_bits = 0;                // bits 1 | 2 of k_step's auto_reset argument: record-first / key-less staging (set at creation)
    bool samples_ready = false;        // k_extract_samples' LDS attribute set
    // move() API state, allocated by the first meshenv_move: not_valid_points per env
    double2 *nv_xy = nullptr;    // [E][cap]
    int32_t *nv_count = nullptr; // [E]
    int32_t *nv_gid = nullptr;   // [E][cap] ring id of each listed vertex (identity, for last_not_valid_points)
    int32_t *nv_meta = nullptr;  // [E][kNvMeta] episode counter + last_not_valid_points summary
    uint8_t *move_mask = nullptr;  // [E] envs of the last meshenv_move that went through smooth_pave
    long long ev_count = 0;      // launches recorded since timing was armed
};

namespace {

#define HIP_TRY(h, expr)                                                                            \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) {                                                                     \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                           \
            return MESHENV_E_HIP;                                                                   \
        }                                                                                           \
    } while (0)

template <typename T>
int dev_alloc(MeshEnv *h, T **out, size_t count)
{
    void *p = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc(&p, count * sizeof(T));
    if (e != hipSuccess) {
        h->err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return MESHENV_E_HIP;
    }
    h->allocs.push_back(p);
    *out = (T *)p;
    return MESHENV_OK;
}

// Every entry point that launches, copies or synchronises runs on the HANDLE's device and leaves the caller's current
// device as it found it (a process may hold handles on several GPUs, and torch tracks its own current device).
struct DeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) err = hipSetDevice(device);
        else prev = -1;  // nothing to restore
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define MESHENV_ON_DEVICE(h)              \
    DeviceGuard _guard((h)->device);      \
    HIP_TRY(h, _guard.err)

int fail_arg(MeshEnv *h, const char *msg)
{
    if (h) h->err = msg;
    else g_create_error = msg;
    return MESHENV_E_ARG;
}

}  // namespace

extern "C" {

void meshenv_default_params(MeshEnvParams *p)
{
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->struct_size = (int32_t)sizeof(MeshEnvParams);
    p->neighbor_num = 6;
    p->radius_num = 3;
    p->fail_limit = 100;
    p->log_capacity = 0;
    p->radius = kDefRadius;
    p->max_ref_angle = kDefMaxRefAngle;
    p->key_lambda = kDefKeyLambda;
    p->min_degree = kDefMinDegree;
    p->max_degree = kDefMaxDegree;
    p->same_point_eps = kDefSameEps;
    p->ray_length = kDefRayLength;
}

int meshenv_abi_version(void) { return MESHENV_ABI_VERSION; }

int meshenv_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *meshenv_last_error(const MeshEnv *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

void meshenv_destroy(MeshEnv *h)
{
    if (!h) return;
    DeviceGuard guard(h->device);
    (void)hipStreamSynchronize(h->stream);
    for (void *p : h->allocs) (void)hipFree(p);
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    delete h;
}

}  // extern "C"

// shared body of meshenv_create (domains from host arrays) and meshenv_create_random (gen != NULL: the rings and their
// constants are produced on the device, csrc/meshenv_domgen.h; dom_xy_host / dom_consts_host are NULL then)
static int create_impl(int device, int n_domains, const int32_t *dom_offsets_host, const double *dom_xy_host,
                       const double *dom_consts_host, int n_envs, const int32_t *env_domain_host,
                       const MeshEnvParams *params, void *stream, const GenParams *gen, MeshEnv **out)
{
    if (!out) return fail_arg(nullptr, "meshenv_create: out is NULL");
    *out = nullptr;
    if (n_domains <= 0 || n_envs <= 0 || !dom_offsets_host || !env_domain_host || (!gen && (!dom_xy_host || !dom_consts_host)))
        return fail_arg(nullptr, "meshenv_create: null or empty input");
    MeshEnvParams prm;
    meshenv_default_params(&prm);
    if (params) {
        if (params->struct_size != (int32_t)sizeof(MeshEnvParams))
            return fail_arg(nullptr, "meshenv_create: MeshEnvParams.struct_size mismatch (ABI)");
        prm = *params;
    }
    if (prm.neighbor_num != 6 || prm.radius_num != 3)
        return fail_arg(nullptr, "meshenv_create: neighbor_num must be 6 and radius_num 3 (observation layout)");
    if (prm.fail_limit <= 0 || prm.log_capacity < 0 || !(prm.radius > 0))
        return fail_arg(nullptr, "meshenv_create: bad parameter value");

    int max_ring = 0;
    for (int d = 0; d < n_domains; d++) {
        const int n0 = dom_offsets_host[d + 1] - dom_offsets_host[d];
        if (n0 < 4) return fail_arg(nullptr, "meshenv_create: a domain ring needs at least 4 vertices");
        max_ring = n0 > max_ring ? n0 : max_ring;
    }
    for (int e = 0; e < n_envs; e++)
        if (env_domain_host[e] < 0 || env_domain_host[e] >= n_domains)
            return fail_arg(nullptr, "meshenv_create: env_domain entry out of range");
    const int cap = (max_ring + 15) / 16 * 16;
    const size_t lds = lds_bytes_for(cap);
    if (lds > 160 * 1024) return fail_arg(nullptr, "meshenv_create: ring too long for one CU's LDS (160 KB: about 3600 vertices)");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_create_error = "meshenv_create: no HIP device available (this library has no CPU fallback)";
        return MESHENV_E_HIP;
    }
    if (device < 0 || device >= ndev) return fail_arg(nullptr, "meshenv_create: device index out of range");

    MeshEnv *h = new MeshEnv();
    h->device = device;
    h->stream = (hipStream_t)stream;
    h->n_envs = n_envs;
    h->n_domains = n_domains;
    h->cap = cap;
    h->max_ring = max_ring;
    h->lds = lds;
#define CREATE_TRY(expr)                                            \
    do {                                                            \
        int _rc = (expr);                                           \
        if (_rc != MESHENV_OK) {                                    \
            g_create_error = h->err;                                \
            meshenv_destroy(h);                                     \
            return _rc;                                             \
        }                                                           \
    } while (0)
#define CREATE_HIP(expr)                                                                \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            g_create_error = std::string(#expr) + ": " + hipGetErrorString(_e);         \
            meshenv_destroy(h);                                                         \
            return MESHENV_E_HIP;                                                       \
        }                                                                               \
    } while (0)

    DeviceGuard guard(device);
    CREATE_HIP(guard.err);
    // The attribute is per function, not per handle: always raise it to the CU's full 160 KB so that a later handle with
    // shorter rings cannot lower the cap under an earlier one.
    constexpr int kLdsCap = 160 * 1024;
    if (lds > 64 * 1024) {
        CREATE_HIP(hipFuncSetAttribute((const void *)k_step<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsCap));
        CREATE_HIP(hipFuncSetAttribute((const void *)k_step<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsCap));
        CREATE_HIP(hipFuncSetAttribute((const void *)k_step<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsCap));
        CREATE_HIP(hipFuncSetAttribute((const void *)k_step<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsCap));
        CREATE_HIP(hipFuncSetAttribute((const void *)k_step<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsCap));
        CREATE_HIP(hipFuncSetAttribute((const void *)k_step<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsCap));
        CREATE_HIP(hipFuncSetAttribute((const void *)k_step<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsCap));
        CREATE_HIP(hipFuncSetAttribute((const void *)k_step<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsCap));
        CREATE_HIP(hipFuncSetAttribute((const void *)k_step<false, true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsCap));
        CREATE_HIP(hipFuncSetAttribute((const void *)k_reset, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsCap));
        CREATE_HIP(hipFuncSetAttribute((const void *)k_init_domains, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsCap));
    }
    CREATE_HIP(upload_atan_state());   // the tie-breaker of every quantised angle (csrc/meshenv_geom.h, atan2_nc)

    // Single-step kernel variant.  The CU-group kernel (G envs per workgroup, SIMD-balanced updates) pays off in the
    // latency-bound regime only: one workgroup per CU (n_envs <= 256 * G) with G >= 8; measured on MI355X,
    // boundary(): 4096 envs 20.2 -> 19.4 us/step (G = 16), 2048 envs 16.5 -> 16.0 (G = 8), but 8192 envs
    // 25.0 -> 34.6 and 65536 envs 103 -> 231 us/step, where throughput, not the slowest wave, sets the time.
    {
        int n_cu = 256;  // MI355X; read from the device so that a partitioned GPU (CPX / fewer CUs) keeps one workgroup per CU
        CREATE_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device));
        if (n_cu <= 0) n_cu = 256;
        h->n_cu = n_cu;
        int G = 1;
        const char *force = getenv("MESHENV_GROUP");
        int want = 1;
        if (force) want = atoi(force);
        else if (n_envs <= n_cu * 16) {
            int g = 1;
            while (g * 2 <= n_envs / n_cu && g * 2 <= 16) g *= 2;
            want = g >= 8 ? g : 1;
        }
        // (G = 4, the speculative kernel and the T-step closed-loop kernel are measured-slower experiments: they exist in the
        // -DMESHENV_DEV build that tools/ and tests/test_gpu_variants.py compile, not in the shipped library)
#ifdef MESHENV_DEV
        for (int g : {16, 8, 4})
#else
        for (int g : {16, 8})
#endif
            if (g <= want && group_lds_bytes(cap, g) <= 150 * 1024) { G = g; break; }
        if (!force && (G < 8 || n_envs > n_cu * G)) G = 1;  // LDS forced a smaller group: more than one workgroup per CU
        // Ragged packing: sixteen rings of the LONGEST stride do not fit, sixteen rings of their own lengths may (mixed
        // d1 / d2 / d3: 120 / 196 / 272 vertices -> 135 KB instead of 197 KB).  One workgroup per CU as in the uniform case.
        if (G == 1 && want == 16 && n_envs <= n_cu * 16 && group_lds_bytes(cap, 16) > 150 * 1024) {
            std::vector<int2> pack((size_t)n_envs);
            size_t worst = 0;
            for (int w0 = 0; w0 < n_envs; w0 += 16) {
                size_t off = 0;
                for (int e = w0; e < w0 + 16 && e < n_envs; e++) {
                    const int d = env_domain_host[e];
                    const int n0 = dom_offsets_host[d + 1] - dom_offsets_host[d];
                    const int cap_e = (n0 + 15) / 16 * 16;
                    pack[(size_t)e] = make_int2((int)off, cap_e);
                    off += lds_bytes_for(cap_e);
                }
                worst = off > worst ? off : worst;
            }
            worst = (worst + 31) / 32 * 32;
            if (worst + 16 * sizeof(Handoff) <= (size_t)kLdsCap) {   // one workgroup per CU: the whole 160 KB may go to it
                G = 16;
                h->env_lds_host.swap(pack);
                h->ho_off = (int)worst;
            }
        }
        MeshEnvParams def;
        meshenv_default_params(&def);
        h->default_params = prm.radius == def.radius && prm.max_ref_angle == def.max_ref_angle && prm.key_lambda == def.key_lambda &&
                            prm.min_degree == def.min_degree && prm.max_degree == def.max_degree &&
                            prm.same_point_eps == def.same_point_eps && prm.ray_length == def.ray_length;
        if (!h->default_params) G = 1;  // the group kernels exist with literal (default) geometry constants only
        if (!h->default_params) h->env_lds_host.clear();
        h->group = G;
        h->group_lds = h->env_lds_host.empty() ? group_lds_bytes(cap, G) : (size_t)h->ho_off + 16 * sizeof(Handoff);
        // The speculative form of the CU-group kernel (k_step_spec: two ring buffers per env, no workgroup barrier) is
        // opt-in, MESHENV_SPEC=1: measured on MI355X at 4096 x boundary() it takes 16.7 us per launch against 15.5 us
        // for the barrier + deal form (rocprofv3; DESIGN.md section 5 says why), so the default stays k_step_group.
#ifdef MESHENV_DEV
        const char *spec_env = getenv("MESHENV_SPEC");
        h->spec = h->env_lds_host.empty() && (G == 16 || G == 8) && spec_lds_bytes(cap, G) <= 150 * 1024 && spec_env && atoi(spec_env) == 1;
#else
        h->spec = false;
#endif
        // staging mode of the one-wave-per-env step kernel (bits 1, 2 of its auto_reset argument), decided once per handle:
        // record-first staging (memoised rejections never load their ring) -- measured never slower from 8192 envs up,
        // +9..15 % at 32 768+; the ring staged without candidate keys / stamps -- +0..3 % from 32 768 envs, -2 % at 8192.
        // MESHENV_LAZY / MESHENV_LIGHT override the size rule (A/B switches, read here, not per launch).
        {
            const char *lz = getenv("MESHENV_LAZY");
            const bool lazy = lz ? atoi(lz) != 0 : n_envs >= 8192;
            const char *lt = getenv("MESHENV_LIGHT");
            const bool light = lt ? atoi(lt) != 0 : (lazy && n_envs >= 16384);
            h->stage_bits = (lazy ? 2 : 0) | (light ? 4 : 0);
        }
#ifdef MESHENV_DEV
        if (h->spec) {
            h->group_lds = spec_lds_bytes(cap, G);
            if (h->group_lds > 64 * 1024) {
                const void *fn = G == 16 ? (const void *)k_step_spec<16, true> : (const void *)k_step_spec<8, true>;
                CREATE_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsCap));
            }
        } else
#endif
        if (G > 1 && h->group_lds > 64 * 1024) {
            const bool ragged = !h->env_lds_host.empty();
            const void *fn = ragged ? (const void *)k_step_group<16, true, true>
                             : G == 16 ? (const void *)k_step_group<16, true> : (const void *)k_step_group<8, true>;
            if (!ragged && cap <= 64) fn = G == 16 ? (const void *)k_step_group<16, true, false, true> : (const void *)k_step_group<8, true, false, true>;
#ifdef MESHENV_DEV
            if (G == 4) fn = (const void *)k_step_group<4, true>;
#endif
            CREATE_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsCap));
        }
    }

    DevState &S = h->S;
    S.n_domains = n_domains;
    S.n_envs = n_envs;
    S.prm.radius = prm.radius;
    S.prm.max_ref_angle = prm.max_ref_angle;
    S.prm.w0 = prm.key_lambda;
    S.prm.w1 = 1 - prm.key_lambda;
    S.prm.min_degree = prm.min_degree;
    S.prm.max_degree = prm.max_degree;
    S.prm.same_eps = prm.same_point_eps;
    S.prm.ray_length = prm.ray_length;
    S.prm.fail_limit = prm.fail_limit;
    S.prm.log_cap = prm.log_capacity;

    const int total_dom = dom_offsets_host[n_domains];
    h->dom_off_host.assign(dom_offsets_host, dom_offsets_host + n_domains + 1);
    h->env_dom_host.assign(env_domain_host, env_domain_host + n_envs);
    const size_t total_env = (size_t)n_envs * (size_t)cap;  // uniform ring stride
    S.cap = cap;

    double2 *d_dom_xy = nullptr;
    DevCold &K = h->cold;  // host copy of the rarely used pointers; the kernels read the device copy S.cold
    std::memset(&K, 0, sizeof(K));
    CREATE_TRY(dev_alloc(h, &S.dom, (size_t)n_domains));
    CREATE_TRY(dev_alloc(h, &d_dom_xy, (size_t)total_dom));
    CREATE_TRY(dev_alloc(h, &K.dom_key, (size_t)total_dom));
    CREATE_TRY(dev_alloc(h, &K.dom_stamp, (size_t)total_dom));
    CREATE_TRY(dev_alloc(h, &K.dom_obs, (size_t)n_domains * kObsDim));
    CREATE_TRY(dev_alloc(h, &S.ring_xy, total_env));
    CREATE_TRY(dev_alloc(h, &S.ring_id, total_env));
    CREATE_TRY(dev_alloc(h, &S.ring_key, total_env));
    CREATE_TRY(dev_alloc(h, &S.ring_stamp, total_env));
    CREATE_TRY(dev_alloc(h, &S.scal, (size_t)n_envs));
    CREATE_TRY(dev_alloc(h, &S.cnt, (size_t)n_envs));
    CREATE_TRY(dev_alloc(h, &S.obs_cache, (size_t)n_envs * kObsDim));
    if (!h->env_lds_host.empty()) {
        CREATE_TRY(dev_alloc(h, &h->env_lds, (size_t)n_envs));
        CREATE_HIP(hipMemcpy(h->env_lds, h->env_lds_host.data(), sizeof(int2) * (size_t)n_envs, hipMemcpyHostToDevice));
    }
    if (prm.log_capacity > 0) {
        CREATE_TRY(dev_alloc(h, &K.log_quads, (size_t)n_envs * 2 * prm.log_capacity * 4));
        CREATE_TRY(dev_alloc(h, &K.log_vxy, (size_t)n_envs * 2 * prm.log_capacity));
        CREATE_TRY(dev_alloc(h, &K.last_ep, (size_t)n_envs));
        CREATE_HIP(hipMemsetAsync(K.last_ep, 0, sizeof(LastEpisode) * (size_t)n_envs, h->stream));
    }
    K.dom_xy = d_dom_xy;
    {
        DevCold *cold_dev = nullptr;
        CREATE_TRY(dev_alloc(h, &cold_dev, (size_t)1));
        CREATE_HIP(hipMemcpy(cold_dev, &K, sizeof(DevCold), hipMemcpyHostToDevice));
        S.cold = cold_dev;
    }
#ifdef MESHENV_STAMPS
    CREATE_TRY(dev_alloc(h, &S.dbg, (size_t)n_envs * 16));
#endif

    std::vector<EnvScalars> sc((size_t)n_envs);
    std::memset(sc.data(), 0, sc.size() * sizeof(EnvScalars));
    for (int e = 0; e < n_envs; e++) sc[e].dom = env_domain_host[e];

    if (!gen) {
        std::vector<DomConst> dc((size_t)n_domains);
        std::memset(dc.data(), 0, dc.size() * sizeof(DomConst));
        for (int d = 0; d < n_domains; d++) {
            dc[d].orig_area = dom_consts_host[3 * d];
            dc[d].min_area = dom_consts_host[3 * d + 1] * dom_consts_host[3 * d + 1];   // estimated_area_range[0] ** 2
            dc[d].crit_area = dom_consts_host[3 * d + 2] * dom_consts_host[3 * d + 2];  // estimated_area_range[1] ** 2
            dc[d].off = dom_offsets_host[d];
            dc[d].n0 = dom_offsets_host[d + 1] - dom_offsets_host[d];
            dc[d].ref = -1;
        }
        CREATE_HIP(hipMemcpy(S.dom, dc.data(), sizeof(DomConst) * (size_t)n_domains, hipMemcpyHostToDevice));
        CREATE_HIP(hipMemcpy(d_dom_xy, dom_xy_host, sizeof(double2) * (size_t)total_dom, hipMemcpyHostToDevice));
    } else {
        // the rings at their offsets, then the constants: two launches, nothing generated on the host
        int32_t *d_off = nullptr;
        int *d_err = nullptr;
        CREATE_TRY(dev_alloc(h, &d_off, (size_t)n_domains + 1));
        CREATE_TRY(dev_alloc(h, &d_err, (size_t)1));
        CREATE_HIP(hipMemcpy(d_off, dom_offsets_host, sizeof(int32_t) * ((size_t)n_domains + 1), hipMemcpyHostToDevice));
        CREATE_HIP(hipMemset(d_err, 0, sizeof(int)));
        if (gen->density_mode)
            hipLaunchKernelGGL(k_gen_rings_density, dim3(n_domains), dim3(64), 0, h->stream, *gen, n_domains, (const int32_t *)d_off, d_dom_xy, d_err);
        else
            hipLaunchKernelGGL(k_gen_rings, dim3(n_domains), dim3(64), 0, h->stream, *gen, n_domains, (const int32_t *)d_off, d_dom_xy, d_err);
        CREATE_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_dom_consts, dim3(n_domains), dim3(64), sizeof(double2) * (size_t)max_ring, h->stream, n_domains,
                           (const int32_t *)d_off, (const double2 *)d_dom_xy, S.dom);
        CREATE_HIP(hipGetLastError());
        int err = 0;
        CREATE_HIP(hipMemcpy(&err, d_err, sizeof(int), hipMemcpyDeviceToHost));
        if (err) {
            g_create_error = "meshenv_create_random: a ring could not be generated on the device (fewer than 5 distinct pixels, or a length mismatch between the two passes)";
            meshenv_destroy(h);
            return MESHENV_E_STATE;
        }
    }
    CREATE_HIP(hipMemcpy(S.scal, sc.data(), sizeof(EnvScalars) * (size_t)n_envs, hipMemcpyHostToDevice));
    CREATE_HIP(hipMemset(S.ring_xy, 0, sizeof(double2) * total_env));
    CREATE_HIP(hipMemset(S.ring_id, 0, sizeof(int32_t) * total_env));
    CREATE_HIP(hipMemset(S.ring_key, 0, sizeof(double) * total_env));
    CREATE_HIP(hipMemset(S.ring_stamp, 0, sizeof(int32_t) * total_env));

    hipLaunchKernelGGL(k_init_domains, dim3(n_domains), dim3(64), lds, h->stream, S, cap);
    CREATE_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_reset, dim3(n_envs), dim3(64), lds, h->stream, S, cap, (const uint8_t *)nullptr, (float *)nullptr, 1, 0ULL,
                       0, (int32_t *)nullptr, (int32_t *)nullptr);
    CREATE_HIP(hipGetLastError());
    CREATE_HIP(hipStreamSynchronize(h->stream));
#undef CREATE_TRY
#undef CREATE_HIP
    *out = h;
    return MESHENV_OK;
}

extern "C" {

int meshenv_create(int device, int n_domains, const int32_t *dom_offsets_host, const double *dom_xy_host,
                   const double *dom_consts_host, int n_envs, const int32_t *env_domain_host,
                   const MeshEnvParams *params, void *stream, MeshEnv **out)
{
    return create_impl(device, n_domains, dom_offsets_host, dom_xy_host, dom_consts_host, n_envs, env_domain_host, params,
                       stream, nullptr, out);
}

// clockwise_angle(a, b) of ui/tk-ui.py:185-192 and its cosine / sine for the offset (dx, dy) = b - a: the host libm's
// values -- the reference's -- in the reference's expression order (domains.clockwise_angle).  integer: the coordinates
// are Python ints, so -(b[1] - a[1]) of a zero difference is the int 0, i.e. +0.0 in atan2 (a float difference gives -0.0).
static void edge_direction(double dx, double dy, bool integer, double *cs)
{
    volatile double ny = (integer && dy == 0) ? 0.0 : -dy, x = dx;   // volatile: the calls must reach the running libm
    const double theta = -std::atan2(ny, x);
    volatile double angle = std::copysign(1.0, theta) >= 0 ? theta : 2 * 3.141592653589793 + theta;
    cs[0] = std::cos(angle);
    cs[1] = std::sin(angle);
}

// shared by meshenv_create_random (uniform split) and meshenv_create_random_density (calculate_density)
static int create_random_impl(const char *fn, int device, int n_envs, GenParams gp, const uint64_t *seeds_host,
                              const MeshEnvParams *params, void *stream, MeshEnv **out, uint8_t *raises_host)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_create_error = std::string(fn) + ": no HIP device available (this library has no CPU fallback)";
        return MESHENV_E_HIP;
    }
    if (device < 0 || device >= ndev) return fail_arg(nullptr, "meshenv_create_random: device index out of range");
    // pass 1: the ring lengths (the domain table and the ring stride are sized from them)
    std::vector<int32_t> offs((size_t)n_envs + 1, 0), env_dom((size_t)n_envs);
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) { g_create_error = std::string(fn) + ": hipSetDevice failed"; return MESHENV_E_HIP; }
    int32_t *d_cnt = nullptr;
    int *d_err = nullptr;
    unsigned long long *d_seeds = nullptr;
    double2 *d_tab = nullptr;
    unsigned char *d_raises = nullptr;
    auto cleanup = [&]() {
        if (d_cnt) (void)hipFree(d_cnt);
        if (d_err) (void)hipFree(d_err);
        if (d_seeds) (void)hipFree(d_seeds);
        if (d_tab) (void)hipFree(d_tab);
        if (d_raises) (void)hipFree(d_raises);
    };
    bool ok = hipMalloc((void **)&d_cnt, sizeof(int32_t) * (size_t)n_envs) == hipSuccess &&
              hipMalloc((void **)&d_err, sizeof(int)) == hipSuccess && hipMemset(d_err, 0, sizeof(int)) == hipSuccess;
    if (ok && seeds_host) {
        ok = hipMalloc((void **)&d_seeds, sizeof(unsigned long long) * (size_t)n_envs) == hipSuccess &&
             hipMemcpy(d_seeds, seeds_host, sizeof(unsigned long long) * (size_t)n_envs, hipMemcpyHostToDevice) == hipSuccess;
        gp.seeds = d_seeds;
    }
    if (ok && gp.density_mode) {
        // pixel coordinates lie in ctr -+ 2 aveRadius (radii are clipped to [0, 2 aveRadius]): offsets within 4 aveRadius + 1
        const int R = (int)std::ceil(4 * gp.ave_radius) + 1, W = 2 * R + 1;
        std::vector<double> tab((size_t)W * W * 2);
        for (int dx = -R; dx <= R; dx++)
            for (int dy = -R; dy <= R; dy++) edge_direction((double)dx, (double)dy, true, &tab[((size_t)(dx + R) * W + (dy + R)) * 2]);
        ok = hipMalloc((void **)&d_tab, sizeof(double2) * (size_t)W * W) == hipSuccess &&
             hipMemcpy(d_tab, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice) == hipSuccess &&
             hipMalloc((void **)&d_raises, (size_t)n_envs) == hipSuccess && hipMemset(d_raises, 0, (size_t)n_envs) == hipSuccess;
        gp.dir_tab = d_tab;
        gp.dir_r = R;
        gp.raises = d_raises;
    }
    int err = 0;
    if (ok) {
        if (gp.density_mode) hipLaunchKernelGGL(k_gen_count_density, dim3(n_envs), dim3(64), 0, (hipStream_t)stream, gp, n_envs, d_cnt, d_err);
        else hipLaunchKernelGGL(k_gen_count, dim3(n_envs), dim3(64), 0, (hipStream_t)stream, gp, n_envs, d_cnt, d_err);
        ok = hipGetLastError() == hipSuccess && hipStreamSynchronize((hipStream_t)stream) == hipSuccess &&
             hipMemcpy(offs.data() + 1, d_cnt, sizeof(int32_t) * (size_t)n_envs, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(&err, d_err, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
    }
    int n_raise = 0;
    if (ok && gp.density_mode) {
        std::vector<uint8_t> rz((size_t)n_envs);
        ok = hipMemcpy(rz.data(), d_raises, (size_t)n_envs, hipMemcpyDeviceToHost) == hipSuccess;
        for (int k = 0; k < n_envs; k++) n_raise += rz[(size_t)k] ? 1 : 0;
        if (raises_host) std::memcpy(raises_host, rz.data(), (size_t)n_envs);
    }
    if (!ok) { cleanup(); g_create_error = std::string(fn) + ": the ring-length pass failed (HIP error)"; return MESHENV_E_HIP; }
    if (err) {
        cleanup();
        g_create_error = std::string(fn) + ": a ring could not be generated on the device (fewer than 5 distinct pixels, or a densified ring beyond 2048 points)";
        return MESHENV_E_STATE;
    }
    if (n_raise) {
        cleanup();
        g_create_error = std::string(fn) + ": " + std::to_string(n_raise) + " of " + std::to_string(n_envs) +
                         " polygons have no ring here -- raises_host[k] = 1: calculate_density raises ZeroDivisionError (an edge of "
                         "0.5 - 1.5 spacings, ui/tk-ui.py:263-264); 2: not generated on the device (fewer than 5 distinct pixels, a "
                         "densified ring beyond 2048 points, an edge outside the direction table) -- pass seeds without them";
        return MESHENV_E_STATE;
    }
    if (!out) { cleanup(); return MESHENV_OK; }   // probe only
    for (int k = 0; k < n_envs; k++) {
        if (offs[(size_t)k + 1] < 4) { cleanup(); g_create_error = std::string(fn) + ": empty ring"; return MESHENV_E_STATE; }
        offs[(size_t)k + 1] += offs[(size_t)k];
        env_dom[(size_t)k] = k;
    }
    gp.raises = nullptr;   // pass 2 runs on rings that do not raise
    const int rc = create_impl(device, n_envs, offs.data(), nullptr, nullptr, n_envs, env_dom.data(), params, stream, &gp, out);
    cleanup();
    return rc;
}

static GenParams default_gen_params(uint64_t seed0, int num_verts)
{
    GenParams gp;
    std::memset(&gp, 0, sizeof(gp));
    gp.seed0 = seed0;
    gp.ctr_x = 250; gp.ctr_y = 250; gp.ave_radius = 100; gp.irregularity = 0.55; gp.spikeyness = 0.7;  // GenerateRandomPolygon.py:63
    gp.fixed_verts = num_verts;
    return gp;
}

int meshenv_create_random(int device, int n_envs, uint64_t seed0, int num_verts, double edge, const MeshEnvParams *params,
                          void *stream, MeshEnv **out)
{
    if (!out) return fail_arg(nullptr, "meshenv_create_random: out is NULL");
    *out = nullptr;
    if (n_envs <= 0 || !(edge > 0.0) || num_verts < 0 || (num_verts > 0 && (num_verts < 5 || num_verts > kGenMaxVerts)))
        return fail_arg(nullptr, "meshenv_create_random: n_envs > 0, edge > 0 and num_verts in {0, 5..64} are required");
    GenParams gp = default_gen_params(seed0, num_verts);
    gp.edge = edge;
    return create_random_impl("meshenv_create_random", device, n_envs, gp, nullptr, params, stream, out, nullptr);
}

int meshenv_create_random_density(int device, int n_envs, uint64_t seed0, const uint64_t *seeds_host, int num_verts,
                                  double base_length, double density, const MeshEnvParams *params, void *stream, MeshEnv **out,
                                  uint8_t *raises_host)
{
    if (out) *out = nullptr;
    if (n_envs <= 0 || !(base_length > 0.0) || !(density > 0.0) || num_verts < 0 ||
        (num_verts > 0 && (num_verts < 5 || num_verts > kGenMaxVerts)))
        return fail_arg(nullptr, "meshenv_create_random_density: n_envs > 0, base_length > 0, density > 0 and num_verts in {0, 5..64} are required");
    GenParams gp = default_gen_params(seed0, num_verts);
    gp.density_mode = 1;
    gp.base_length = base_length;
    gp.density = density;
    return create_random_impl("meshenv_create_random_density", device, n_envs, gp, seeds_host, params, stream, out, raises_host);
}

int meshenv_density_rings(int device, int n_polys, const int32_t *poly_offsets_host, const double *pixels_host, int integer_pixels,
                          const double *densities_host, double base_length, int32_t *count_host, uint8_t *status_host,
                          double *xy_host, int64_t cap_points)
{
    if (n_polys <= 0 || !poly_offsets_host || !pixels_host || !count_host || !status_host || !(base_length > 0.0))
        return MESHENV_E_ARG;
    const int total_in = poly_offsets_host[n_polys];
    for (int k = 0; k < n_polys; k++) {
        const int nv = poly_offsets_host[k + 1] - poly_offsets_host[k];
        if (nv < 3 || nv > kDensMaxVerts) return MESHENV_E_ARG;
    }
    if (integer_pixels)
        for (int i = 0; i < 2 * total_in; i++)
            if (pixels_host[i] != std::floor(pixels_host[i]) || std::fabs(pixels_host[i]) > 1e9) return MESHENV_E_ARG;
    // Edge directions and lengths from the host libm.  The device deduplicates repeated pixels (the reference's dict), so
    // the edge INTO deduplicated vertex r is computed here for the same deduplicated list.
    std::vector<double> dir((size_t)total_in * 2, 0.0), len((size_t)total_in, 0.0);
    volatile double two = 2.0;
    for (int k = 0; k < n_polys; k++) {
        const int o = poly_offsets_host[k], nv = poly_offsets_host[k + 1] - o;
        const double *P = pixels_host + 2 * (size_t)o;
        std::vector<int> keep;
        for (int i = 0; i < nv; i++) {
            bool first = true;
            for (int j = 0; j < i && first; j++) first = !(P[2 * j] == P[2 * i] && P[2 * j + 1] == P[2 * i + 1]);
            if (first) keep.push_back(i);
        }
        const int m = (int)keep.size();
        for (int r = 0; r < m; r++) {
            const int i = keep[(size_t)r], ip = keep[(size_t)(r == 0 ? m - 1 : r - 1)];
            const double dx = P[2 * i] - P[2 * ip], dy = P[2 * i + 1] - P[2 * ip + 1];
            edge_direction(dx, dy, integer_pixels != 0, &dir[2 * (size_t)(o + r)]);
            // distance(): math.sqrt((p1[0] - p2[0]) ** 2 + (p1[1] - p2[1]) ** 2) -- exact int arithmetic for Python ints,
            // libm pow for floats
            len[(size_t)(o + r)] = integer_pixels ? std::sqrt(dx * dx + dy * dy) : std::sqrt(std::pow(-dx, two) + std::pow(-dy, two));
        }
    }
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return MESHENV_E_HIP;
    int32_t *d_off = nullptr, *d_cnt = nullptr, *d_ooff = nullptr;
    double *d_dens = nullptr, *d_px = nullptr, *d_len = nullptr;
    double2 *d_dir = nullptr, *d_out = nullptr;
    unsigned char *d_st = nullptr;
    int rc = MESHENV_OK;
    auto H = [&](hipError_t e) { if (e != hipSuccess && rc == MESHENV_OK) rc = MESHENV_E_HIP; return e == hipSuccess; };
    H(hipMalloc((void **)&d_off, sizeof(int32_t) * ((size_t)n_polys + 1)));
    H(hipMalloc((void **)&d_px, sizeof(double) * 2 * (size_t)total_in));
    H(hipMalloc((void **)&d_cnt, sizeof(int32_t) * (size_t)n_polys));
    H(hipMalloc((void **)&d_ooff, sizeof(int32_t) * ((size_t)n_polys + 1)));
    H(hipMalloc((void **)&d_dir, sizeof(double2) * (size_t)total_in));
    H(hipMalloc((void **)&d_len, sizeof(double) * (size_t)total_in));
    H(hipMalloc((void **)&d_st, (size_t)n_polys));
    if (densities_host) H(hipMalloc((void **)&d_dens, sizeof(double) * (size_t)total_in));
    if (rc == MESHENV_OK) {
        H(hipMemcpy(d_off, poly_offsets_host, sizeof(int32_t) * ((size_t)n_polys + 1), hipMemcpyHostToDevice));
        H(hipMemcpy(d_px, pixels_host, sizeof(double) * 2 * (size_t)total_in, hipMemcpyHostToDevice));
        H(hipMemcpy(d_dir, dir.data(), sizeof(double) * dir.size(), hipMemcpyHostToDevice));
        H(hipMemcpy(d_len, len.data(), sizeof(double) * len.size(), hipMemcpyHostToDevice));
        if (densities_host) H(hipMemcpy(d_dens, densities_host, sizeof(double) * (size_t)total_in, hipMemcpyHostToDevice));
    }
    std::vector<int32_t> ooff((size_t)n_polys + 1, 0);
    if (rc == MESHENV_OK) {
        hipLaunchKernelGGL(k_density_rings<false>, dim3(n_polys), dim3(64), 0, nullptr, n_polys, (const int32_t *)d_off, (const double *)d_px,
                           (const double *)d_dens, (const double2 *)d_dir, (const double *)d_len, base_length, (const int32_t *)nullptr,
                           (double2 *)nullptr, d_cnt, d_st);
        H(hipGetLastError());
        H(hipDeviceSynchronize());
        H(hipMemcpy(count_host, d_cnt, sizeof(int32_t) * (size_t)n_polys, hipMemcpyDeviceToHost));
        H(hipMemcpy(status_host, d_st, (size_t)n_polys, hipMemcpyDeviceToHost));
    }
    if (rc == MESHENV_OK && xy_host) {
        for (int k = 0; k < n_polys; k++) ooff[(size_t)k + 1] = ooff[(size_t)k] + (status_host[k] == 0 ? count_host[k] : 0);
        const int64_t total_out = ooff[(size_t)n_polys];
        if (total_out > cap_points) rc = MESHENV_E_RANGE;
        else if (total_out > 0) {
            H(hipMalloc((void **)&d_out, sizeof(double2) * (size_t)total_out));
            H(hipMemcpy(d_ooff, ooff.data(), sizeof(int32_t) * ((size_t)n_polys + 1), hipMemcpyHostToDevice));
            if (rc == MESHENV_OK) {
                hipLaunchKernelGGL(k_density_rings<true>, dim3(n_polys), dim3(64), 0, nullptr, n_polys, (const int32_t *)d_off, (const double *)d_px,
                                   (const double *)d_dens, (const double2 *)d_dir, (const double *)d_len, base_length, (const int32_t *)d_ooff,
                                   d_out, d_cnt, d_st);
                H(hipGetLastError());
                H(hipDeviceSynchronize());
                H(hipMemcpy(xy_host, d_out, sizeof(double2) * (size_t)total_out, hipMemcpyDeviceToHost));
            }
        }
    }
    for (void *p : {(void *)d_off, (void *)d_px, (void *)d_cnt, (void *)d_ooff, (void *)d_dens, (void *)d_dir, (void *)d_len, (void *)d_out, (void *)d_st})
        if (p) (void)hipFree(p);
    return rc;
}

int meshenv_get_domain(MeshEnv *h, int domain, double *xy_host, int cap_points, int32_t *n_out, double *consts_host)
{
    if (!h || !n_out) return MESHENV_E_ARG;
    if (domain < 0 || domain >= h->n_domains) {
        h->err = "meshenv_get_domain: domain out of range";
        return MESHENV_E_RANGE;
    }
    MESHENV_ON_DEVICE(h);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const int off = h->dom_off_host[(size_t)domain], n = h->dom_off_host[(size_t)domain + 1] - off;
    *n_out = n;
    if (xy_host) {
        const int m = n < cap_points ? n : cap_points;
        HIP_TRY(h, hipMemcpy(xy_host, h->cold.dom_xy + off, sizeof(double2) * (size_t)m, hipMemcpyDeviceToHost));
    }
    if (consts_host) {
        DomConst dc;
        HIP_TRY(h, hipMemcpy(&dc, h->S.dom + domain, sizeof(dc), hipMemcpyDeviceToHost));
        consts_host[0] = dc.orig_area;
        consts_host[1] = dc.min_area;    // estimated_area_range[0] ** 2
        consts_host[2] = dc.crit_area;   // estimated_area_range[1] ** 2
    }
    return MESHENV_OK;
}

int meshenv_set_stream(MeshEnv *h, void *stream)
{
    if (!h) return MESHENV_E_ARG;
    MESHENV_ON_DEVICE(h);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->stream = (hipStream_t)stream;
    return MESHENV_OK;
}

int meshenv_set_packed_output(MeshEnv *h, float *msg_dev)
{
    if (!h) return MESHENV_E_ARG;
    h->S.msg = msg_dev;
    return MESHENV_OK;
}

int meshenv_num_envs(const MeshEnv *h) { return h ? h->n_envs : MESHENV_E_ARG; }
int meshenv_max_ring(const MeshEnv *h) { return h ? h->max_ring : MESHENV_E_ARG; }
int meshenv_group_size(const MeshEnv *h) { return h ? h->group : MESHENV_E_ARG; }
int meshenv_step_kernel(const MeshEnv *h)
{
    if (!h) return MESHENV_E_ARG;
    if (h->front_moved) return 3;
    if (h->group <= 1) {
        const bool pre = h->default_params && (h->stage_bits & 2);
        return h->default_params && h->cap <= 64 ? (pre ? 8 : 6) : (pre ? 7 : 0);
    }
    if (h->spec) return 2;
    if (h->env_lds) return 4;
#ifdef MESHENV_DEV
    if (h->group == 4) return 1;
#endif
    return h->cap <= 64 ? 5 : 1;
}
int meshenv_libm_exact(const MeshEnv *h) { return h ? h->libm_exact : MESHENV_E_ARG; }
int meshenv_atan2_exact(void) { return atan_host().mode == 2 ? 1 : 0; }

int meshenv_reset_static(MeshEnv *h, const uint8_t *mask_dev, float *obs_dev, int is_static)
{
    if (!h) return MESHENV_E_ARG;
    MESHENV_ON_DEVICE(h);
    hipLaunchKernelGGL(k_reset, dim3(h->n_envs), dim3(64), h->lds, h->stream, h->S, h->cap, mask_dev, obs_dev, 0,
                       (unsigned long long)h->steps_done, is_static ? 1 : 0, h->nv_count, h->nv_meta);
    HIP_TRY(h, hipGetLastError());
    if (!mask_dev) h->front_moved = false;   // every ring is its domain's again: multiples of 1e-4 only
    return MESHENV_OK;
}

int meshenv_reset(MeshEnv *h, const uint8_t *mask_dev, float *obs_dev) { return meshenv_reset_static(h, mask_dev, obs_dev, 0); }

// The front smoother's tan / cos values (csrc/meshenv_smooth.h, "kFt..."): every argument derives from a clockwise angle
// quantised to 1e-4 rad or from a literal, so the host evaluates them all once with ITS libm -- the one the reference's
// math.tan / math.cos call -- in exactly the reference's expression order (math.radians(x) = x * (pi / 180),
// math.degrees(x) = x * (180 / pi); general/mesh.py:809, 839, 872, 886, 953-972, 1048).
static void fill_front_tables(std::vector<double> &t)
{
    // volatile: the calls below must reach the libm of the running process, not be folded by the compiler
    volatile double pi_v = 3.141592653589793;
    const double pi = pi_v, to_rad = pi / 180.0, to_deg = 180.0 / pi;
    t.assign(kFtTotal, 0.0);
    for (int q = 0; q < kFtQ; q++) {
        const double v_angle = ((double)q / 1e4) * to_deg;
        t[kFtCosInd + q] = std::cos(((360 - v_angle) / 2) * to_rad);
    }
    t[kFtTan45] = std::tan(45.0 * to_rad);
    for (int k = 0; k < 10; k++) t[kFtCosSide + k] = std::cos((45.0 - 5.0 * k) * to_rad);
    for (int row = 0; row <= kFtMidQ1 - kFtMidQ0 + 1; row++) {
        double target = row == 0 ? 45.0 : ((double)(kFtMidQ0 + row - 1) / 1e4) * to_deg;
        for (int k = 0; k < kFtMidSteps; k++) {
            t[kFtTanMid + row * kFtMidSteps + k] = std::tan((target / 2) * to_rad);
            target += 5;
        }
    }
    // get_radius_points' bisector (general/components.py:1227-1237): math.cos / math.sin of theta / 2 and of the rotation angle
    for (int q = 0; q < kFtQ; q++) {
        const double a = (double)q / 1e4;
        t[kFtSinCosFull + 2 * q] = std::sin(a);
        t[kFtSinCosFull + 2 * q + 1] = std::cos(a);
        t[kFtSinCosHalf + 2 * q] = std::sin(a / 2);
        t[kFtSinCosHalf + 2 * q + 1] = std::cos(a / 2);
    }
}

// csrc/meshenv_libm.h against the libm of this process: 2^18 arguments m * 2^e, m in [1, 2), e in [-40, 24], both signs
// (the range of coordinate differences and of the quadratic formulas' terms), plus the exact powers of two.
static int validate_pow2()
{
    volatile double two = 2.0;   // keeps the compiler from turning pow(x, 2) into x * x
    unsigned long long s = 88172645463325252ULL;
    for (int n = 0; n < (1 << 18); n++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double m = 1.0 + (double)(s >> 12) / 4503599627370496.0;
        double x = std::ldexp(n < 128 ? 1.0 : m, (int)((s >> 3) % 65) - 40);
        if (s & 1) x = -x;
        if (std::pow(x, two) != pow2_glibc(x)) return 0;
    }
    return 1;
}

// the host-libm table and the pow2 validation, on first use (the smoothing entry points and meshenv_move)
static int ensure_libm_tables(MeshEnv *h)
{
    if (h->front_tab) return MESHENV_OK;
    MESHENV_ON_DEVICE(h);
    if (h->libm_exact < 0) {
        const char *off = std::getenv("MESHENV_LIBM_EXACT");   // "0": square exactly (x * x) whatever the libm does
        h->libm_exact = (off && off[0] == '0') ? 0 : validate_pow2();
    }
    double *tab_dev = nullptr;
    const int rc = dev_alloc(h, &tab_dev, (size_t)kFtTotal);
    if (rc != MESHENV_OK) return rc;
    std::vector<double> tab;
    fill_front_tables(tab);
    // synchronous copy from pageable memory: the vector goes out of scope right after
    HIP_TRY(h, hipMemcpy(tab_dev, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice));
    h->front_tab = tab_dev;   // only now: a failed copy leaves "no table" and the next call tries again
    return MESHENV_OK;
}

// buffers and kernel attributes of the smoothing entry points, on first use
static int ensure_smooth_state(MeshEnv *h)
{
    if (h->smooth_ready) return MESHENV_OK;
    MESHENV_ON_DEVICE(h);
    int rc = ensure_libm_tables(h);
    if (rc != MESHENV_OK) return rc;
    if (!h->smooth_sweeps) rc = dev_alloc(h, &h->smooth_sweeps, (size_t)h->n_envs);
    if (rc != MESHENV_OK) return rc;
    if (!h->front_code) rc = dev_alloc(h, &h->front_code, (size_t)h->n_envs);
    if (rc != MESHENV_OK) return rc;
    HIP_TRY(h, hipFuncSetAttribute((const void *)k_smooth_interior, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(h, hipFuncSetAttribute((const void *)k_smooth_front, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(h, hipFuncSetAttribute((const void *)k_rebuild_candidates<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(h, hipFuncSetAttribute((const void *)k_rebuild_candidates<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(h, hipFuncSetAttribute((const void *)k_rebuild_candidates<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (!h->pend) rc = dev_alloc(h, &h->pend, (size_t)h->n_envs);
    if (rc != MESHENV_OK) return rc;
    if (!h->pend_obs) rc = dev_alloc(h, &h->pend_obs, (size_t)h->n_envs * kObsDim);
    if (rc != MESHENV_OK) return rc;
    HIP_TRY(h, hipMemsetAsync(h->pend, 0xff, sizeof(Reselect) * (size_t)h->n_envs, h->stream));   // n_elem = -1: nothing parked
    h->smooth_ready = true;   // only now: a failure above leaves the state "not ready" and the next call tries again
    return MESHENV_OK;
}

static LibmRef libm_ref(const MeshEnv *h)
{
    LibmRef lr;
    lr.tab = h->front_tab;   // nullptr until ensure_smooth_state ran: find_next_state then evaluates with ocml
    lr.exact = h->libm_exact > 0 ? 1 : 0;
    return lr;
}

// smooth_pave(interior=False) + find_next_state(static = is_static) for the envs of `mask_dev`; sw: [E] outcome
static int launch_full_smoothing(MeshEnv *h, const uint8_t *mask_dev, int iteration, int is_static, int32_t *sw, double *diff_dev,
                                 float *obs_dev)
{
    MESHENV_ON_DEVICE(h);
    const int log_cap = h->S.prm.log_cap;
    const dim3 grid(h->n_envs), block(64);
    h->front_moved = true;   // until the next full reset (launch_step)
    hipLaunchKernelGGL(k_smooth_front, grid, block, smooth_front_lds_bytes(h->cap, log_cap), h->stream, h->S, h->cap, mask_dev,
                       h->front_code, (const double *)h->front_tab, h->libm_exact, h->move_ready ? h->nv_xy : nullptr,
                       (const int32_t *)h->nv_count, (const int32_t *)h->nv_gid);
    HIP_TRY(h, hipGetLastError());
    hipLaunchKernelGGL(k_smooth_interior, grid, block, smooth_lds_bytes(h->cap, log_cap), h->stream, h->S, h->cap, 0, mask_dev,
                       h->front_code, iteration, sw, diff_dev);
    HIP_TRY(h, hipGetLastError());
    if (is_static)
        hipLaunchKernelGGL(k_rebuild_candidates<2>, grid, block, h->lds, h->stream, h->S, h->cap, mask_dev, sw, h->pend, h->pend_obs,
                           obs_dev, libm_ref(h));
    else
        hipLaunchKernelGGL(k_rebuild_candidates<1>, grid, block, h->lds, h->stream, h->S, h->cap, mask_dev, sw, h->pend, h->pend_obs,
                           obs_dev, libm_ref(h));
    HIP_TRY(h, hipGetLastError());
    return MESHENV_OK;
}

static bool smoothing_fits(const MeshEnv *h, bool with_front)
{
    const int log_cap = h->S.prm.log_cap;
    if (log_cap <= 0 || h->cap + log_cap > 65535) return false;
    if (smooth_lds_bytes(h->cap, log_cap) > 160 * 1024) return false;
    return !with_front || smooth_front_lds_bytes(h->cap, log_cap) <= 160 * 1024;
}

static int ensure_move_state(MeshEnv *h)
{
    if (h->move_ready) return MESHENV_OK;
    const size_t total = (size_t)h->n_envs * (size_t)h->cap;
    int rc = MESHENV_OK;
    if (!h->nv_xy) rc = dev_alloc(h, &h->nv_xy, total);
    if (rc != MESHENV_OK) return rc;
    if (!h->nv_count) rc = dev_alloc(h, &h->nv_count, (size_t)h->n_envs);
    if (rc != MESHENV_OK) return rc;
    HIP_TRY(h, hipMemsetAsync(h->nv_count, 0, sizeof(int32_t) * (size_t)h->n_envs, h->stream));
    if (!h->nv_gid) rc = dev_alloc(h, &h->nv_gid, total);
    if (rc != MESHENV_OK) return rc;
    if (!h->nv_meta) rc = dev_alloc(h, &h->nv_meta, (size_t)h->n_envs * kNvMeta);
    if (rc != MESHENV_OK) return rc;
    HIP_TRY(h, hipMemsetAsync(h->nv_meta, 0, sizeof(int32_t) * (size_t)h->n_envs * kNvMeta, h->stream));
    if (!h->move_mask) rc = dev_alloc(h, &h->move_mask, (size_t)h->n_envs);
    if (rc != MESHENV_OK) return rc;
    if (move_lds_bytes(h->cap) > 64 * 1024)
        HIP_TRY(h, hipFuncSetAttribute((const void *)k_move, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    h->move_ready = true;   // only now (see ensure_smooth_state)
    return MESHENV_OK;
}

int meshenv_move(MeshEnv *h, const double *points_dev, const double *type_dev, float *obs_dev, uint8_t *done_dev,
                 uint8_t *complete_dev, uint8_t *code_dev)
{
    if (!h) return MESHENV_E_ARG;
    if (!points_dev || !type_dev || !obs_dev || !done_dev || !complete_dev || !code_dev) return fail_arg(h, "meshenv_move: null device pointer");
    if (move_lds_bytes(h->cap) > 160 * 1024) return fail_arg(h, "meshenv_move: ring too long for the move kernel's LDS (60 B per vertex)");
    MESHENV_ON_DEVICE(h);
    int rc = ensure_move_state(h);
    if (rc == MESHENV_OK) rc = ensure_libm_tables(h);   // before the first k_move: every move computes its observation alike
    if (rc != MESHENV_OK) return rc;
    hipLaunchKernelGGL(k_move, dim3(h->n_envs), dim3(64), move_lds_bytes(h->cap), h->stream, h->S, h->cap, points_dev, type_dev,
                       obs_dev, done_dev, complete_dev, code_dev, h->nv_xy, h->nv_count, h->nv_gid, libm_ref(h));
    HIP_TRY(h, hipGetLastError());
    if (smoothing_fits(h, true)) {
        // B:405-426: the envs k_move left at "no selectable reference vertex" go through smooth_pave and select again
        const int rcs = ensure_smooth_state(h);
        if (rcs != MESHENV_OK) return rcs;
        const int nb = (h->n_envs + 255) / 256;
        hipLaunchKernelGGL(k_move_mask, dim3(nb), dim3(256), 0, h->stream, code_dev, h->move_mask, h->n_envs);
        HIP_TRY(h, hipGetLastError());
        const int rcf = launch_full_smoothing(h, h->move_mask, 400, 1, h->smooth_sweeps, nullptr, obs_dev);
        if (rcf != MESHENV_OK) return rcf;
        hipLaunchKernelGGL(k_move_finish, dim3(nb), dim3(256), 0, h->stream, h->S, h->cap, h->move_mask, h->smooth_sweeps, h->nv_count,
                           h->nv_gid, h->nv_meta, done_dev, complete_dev, code_dev);
        HIP_TRY(h, hipGetLastError());
    }
    if (h->reselect_pending) {  // move() ends with its own selection from the list, accepted or not: nothing stays parked
        HIP_TRY(h, hipMemsetAsync(h->pend, 0xff, sizeof(Reselect) * (size_t)h->n_envs, h->stream));
        h->reselect_pending = false;
    }
    return MESHENV_OK;
}

int meshenv_smooth(MeshEnv *h, int which, const uint8_t *mask_dev, int iteration, int interior, int is_static, int32_t *sweeps_dev,
                   double *diff_dev, float *obs_dev)
{
    if (!h || (which != 0 && which != 1)) return MESHENV_E_ARG;
    if (iteration < 0) return fail_arg(h, "meshenv_smooth: iteration must be >= 0");
    if (which && !interior)
        return fail_arg(h, "meshenv_smooth: the archived episode (which = 1) has no front to step on: interior must be != 0");
    const int log_cap = h->S.prm.log_cap;
    if (log_cap <= 0) {
        h->err = "meshenv_smooth: handle was created with log_capacity = 0 (the mesh graph is rebuilt from the element log)";
        return MESHENV_E_STATE;
    }
    const size_t lds = smooth_lds_bytes(h->cap, log_cap), lds_front = smooth_front_lds_bytes(h->cap, log_cap);
    if (lds > 160 * 1024 || (!interior && lds_front > 160 * 1024) || h->cap + log_cap > 65535)
        return fail_arg(h, "meshenv_smooth: ring stride + log_capacity too large for the smoother's LDS (16 B per ring slot + 60 B per logged vertex; front smoother 51 B per vertex)");
    MESHENV_ON_DEVICE(h);
    {
        const int rc = ensure_smooth_state(h);
        if (rc != MESHENV_OK) return rc;
    }
    int32_t *sw = sweeps_dev ? sweeps_dev : h->smooth_sweeps;
    if (!interior) return launch_full_smoothing(h, mask_dev, iteration, is_static, sw, diff_dev, obs_dev);
    const dim3 grid(h->n_envs), block(64);
    hipLaunchKernelGGL(k_smooth_interior, grid, block, lds, h->stream, h->S, h->cap, which, mask_dev, (const int32_t *)nullptr,
                       iteration, sw, diff_dev);
    HIP_TRY(h, hipGetLastError());
    if (which) return MESHENV_OK;   // an archived mesh: nothing to step on, no candidate list
    hipLaunchKernelGGL(k_rebuild_candidates<0>, grid, block, h->lds, h->stream, h->S, h->cap, mask_dev, sw, h->pend, h->pend_obs,
                       (float *)nullptr, libm_ref(h));
    HIP_TRY(h, hipGetLastError());
    h->reselect_pending = true;
    return MESHENV_OK;
}

int meshenv_smooth_final(MeshEnv *h, int which, const uint8_t *mask_dev, int iteration, double lr_1, double lr_2,
                         int32_t *sweeps_dev, double *diff_dev)
{
    if (!h || (which != 0 && which != 1)) return MESHENV_E_ARG;
    if (iteration < 0) return fail_arg(h, "meshenv_smooth_final: iteration must be >= 0");
    const int log_cap = h->S.prm.log_cap;
    if (log_cap <= 0) {
        h->err = "meshenv_smooth_final: handle was created with log_capacity = 0 (the mesh graph is rebuilt from the element log)";
        return MESHENV_E_STATE;
    }
    const size_t lds = smooth_final_lds_bytes(h->cap, log_cap);
    if (lds > 160 * 1024 || h->cap + log_cap > 65535)
        return fail_arg(h, "meshenv_smooth_final: ring stride + log_capacity too large for the smoother's LDS (51 B per vertex)");
    MESHENV_ON_DEVICE(h);
    if (!h->smooth_final_ready) {
        HIP_TRY(h, hipFuncSetAttribute((const void *)k_smooth_final, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        h->smooth_final_ready = true;
    }
    hipLaunchKernelGGL(k_smooth_final, dim3(h->n_envs), dim3(64), lds, h->stream, h->S, h->cap, which, mask_dev, iteration, lr_1,
                       lr_2, sweeps_dev, diff_dev);
    HIP_TRY(h, hipGetLastError());
    return MESHENV_OK;
}

int meshenv_get_not_valid(MeshEnv *h, int env, double *xy_host, int cap_points, int32_t *count)
{
    if (!h || !count) return MESHENV_E_ARG;
    if (env < 0 || env >= h->n_envs) {
        h->err = "meshenv_get_not_valid: env out of range";
        return MESHENV_E_RANGE;
    }
    *count = 0;
    if (!h->move_ready) return MESHENV_OK;  // move() never called: the list is empty
    MESHENV_ON_DEVICE(h);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    int32_t n = 0;
    HIP_TRY(h, hipMemcpy(&n, h->nv_count + env, sizeof(n), hipMemcpyDeviceToHost));
    *count = n;
    if (xy_host && n > 0) {
        const int m = n < cap_points ? n : cap_points;
        HIP_TRY(h, hipMemcpy(xy_host, h->nv_xy + (size_t)env * h->cap, sizeof(double2) * (size_t)m, hipMemcpyDeviceToHost));
    }
    return MESHENV_OK;
}

int meshenv_get_not_valid_ids(MeshEnv *h, int env, int32_t *ids_host, int cap_ids, int32_t *count, int32_t *last_host)
{
    if (!h || !count) return MESHENV_E_ARG;
    if (env < 0 || env >= h->n_envs) {
        h->err = "meshenv_get_not_valid_ids: env out of range";
        return MESHENV_E_RANGE;
    }
    *count = 0;
    if (last_host) last_host[0] = last_host[1] = last_host[2] = last_host[3] = 0;
    if (!h->move_ready) return MESHENV_OK;  // move() never called
    MESHENV_ON_DEVICE(h);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    int32_t n = 0, meta[kNvMeta];
    HIP_TRY(h, hipMemcpy(&n, h->nv_count + env, sizeof(n), hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(meta, h->nv_meta + (size_t)env * kNvMeta, sizeof(meta), hipMemcpyDeviceToHost));
    EnvScalars s;
    HIP_TRY(h, hipMemcpy(&s, h->S.scal + env, sizeof(s), hipMemcpyDeviceToHost));
    const int n0 = h->dom_off_host[s.dom + 1] - h->dom_off_host[s.dom];
    auto gid = [n0](int32_t g) { return (g & kNewBit) ? n0 + (g & ~kNewBit) : g; };
    *count = n;
    if (ids_host && n > 0) {
        const int m = n < cap_ids ? n : cap_ids;
        HIP_TRY(h, hipMemcpy(ids_host, h->nv_gid + (size_t)env * h->cap, sizeof(int32_t) * (size_t)m, hipMemcpyDeviceToHost));
        for (int i = 0; i < m; i++) ids_host[i] = gid(ids_host[i]);
    }
    if (last_host) {
        last_host[0] = gid(meta[1]); last_host[1] = gid(meta[2]); last_host[2] = meta[3];
        last_host[3] = meta[4] == meta[0] ? 1 : 0;
    }
    return MESHENV_OK;
}

static int launch_step(MeshEnv *h, int n_steps, const float *actions_dev, float *obs_dev, double *reward_dev,
                       uint8_t *done_dev, uint8_t *complete_dev, float *terminal_obs_dev, int auto_reset)
{
    if (!h) return MESHENV_E_ARG;
    if (!actions_dev || !obs_dev || !reward_dev || !done_dev || !complete_dev) return fail_arg(h, "meshenv_step: null device pointer");
    if (n_steps <= 0) return fail_arg(h, "meshenv_rollout: n_steps must be positive");
    if (n_steps > 1 && h->reselect_pending) {
        h->err = "meshenv_rollout: the first step after meshenv_smooth must be a meshenv_step (it commits the re-selection "
                 "the candidate rebuild parked, include/meshenv.h)";
        return MESHENV_E_STATE;
    }
    MESHENV_ON_DEVICE(h);
    const size_t slot = (size_t)(h->ev_count % MESHENV_TIMING_POOL);
    const long long pos = h->timing > 0 ? (h->launch_count++ % (2LL * h->timing)) : -1;
    if (pos == 0) HIP_TRY(h, hipEventRecord(h->ev[2 * slot], h->stream));
    // After a front smoothing (meshenv_smooth(interior = 0), or a meshenv_move that went through smooth_pave) a ring can hold
    // vertices off the 1e-4 lattice, and with them clockwise angles that sit exactly on a rounding boundary: until every env
    // has been reset, steps run the one-wave-per-env kernel in its tie-breaking instantiation (csrc/meshenv_geom.h).
    if (n_steps == 1 && h->group > 1 && !h->front_moved) {
        const int G = h->group;
        const dim3 grid((h->n_envs + G - 1) / G), block(64 * G);
        GroupArgs A;
        A.S = h->S;
        A.outs.obs_out = obs_dev; A.outs.reward = reward_dev; A.outs.done = done_dev; A.outs.complete = complete_dev;
        A.outs.term_obs = terminal_obs_dev;
        A.actions = actions_dev;
        A.step0 = (unsigned long long)h->steps_done;
        A.cap = h->cap;
        A.auto_reset = auto_reset;
        A.env_lds = h->env_lds;
        A.ho_off = h->ho_off;
        A.pad = 0;
#ifdef MESHENV_DEV
        if (h->spec && G == 16) hipLaunchKernelGGL((k_step_spec<16, true>), grid, block, h->group_lds, h->stream, A);
        else if (h->spec && G == 8) hipLaunchKernelGGL((k_step_spec<8, true>), grid, block, h->group_lds, h->stream, A);
        else if (G == 4) hipLaunchKernelGGL((k_step_group<4, true>), grid, block, h->group_lds, h->stream, A);
        else
#endif
        if (h->env_lds) hipLaunchKernelGGL((k_step_group<16, true, true>), grid, block, h->group_lds, h->stream, A);   // ragged: G == 16 only
        else if (h->cap <= 64 && G == 16) hipLaunchKernelGGL((k_step_group<16, true, false, true>), grid, block, h->group_lds, h->stream, A);
        else if (h->cap <= 64) hipLaunchKernelGGL((k_step_group<8, true, false, true>), grid, block, h->group_lds, h->stream, A);
        else if (G == 16) hipLaunchKernelGGL((k_step_group<16, true>), grid, block, h->group_lds, h->stream, A);
        else hipLaunchKernelGGL((k_step_group<8, true>), grid, block, h->group_lds, h->stream, A);
    } else {
        const dim3 grid(h->n_envs), block(64);
#define MESHENV_LAUNCH_STEP(MULTI, DEF)                                                                                      \
    do {                                                                                                                     \
        KStepArgs ka;                                                                                                        \
        ka.S = h->S; ka.cap = h->cap; ka.n_steps = n_steps; ka.actions = actions_dev; ka.obs_out = obs_dev;                  \
        ka.reward = reward_dev; ka.done = done_dev; ka.complete = complete_dev; ka.term_obs = terminal_obs_dev;              \
        ka.auto_reset = auto_reset; ka.step0 = (unsigned long long)h->steps_done;                                            \
        if (h->front_moved) hipLaunchKernelGGL((k_step<MULTI, DEF, true>), grid, block, h->lds, h->stream, ka);              \
        else if (DEF && !MULTI && (h->stage_bits & 2) && h->cap <= 64)                                                       \
            hipLaunchKernelGGL((k_step<false, DEF, false, DEF, DEF>), grid, block, h->lds, h->stream, ka);                   \
        else if (DEF && !MULTI && (h->stage_bits & 2))                                                                       \
            hipLaunchKernelGGL((k_step<false, DEF, false, false, DEF>), grid, block, h->lds, h->stream, ka);                 \
        else if (DEF && h->cap <= 64) hipLaunchKernelGGL((k_step<MULTI, DEF, false, DEF>), grid, block, h->lds, h->stream, ka); \
        else hipLaunchKernelGGL((k_step<MULTI, DEF>), grid, block, h->lds, h->stream, ka);                                   \
    } while (0)
        if (n_steps == 1) {
            // bit 1: record-first staging, bit 2: key-less staging -- throughput regime only, decided in meshenv_create
            auto_reset = (auto_reset ? 1 : 0) | h->stage_bits;
            if (h->default_params) MESHENV_LAUNCH_STEP(false, true);
            else MESHENV_LAUNCH_STEP(false, false);
        } else {
            if (h->default_params) MESHENV_LAUNCH_STEP(true, true);
            else MESHENV_LAUNCH_STEP(true, false);
        }
#undef MESHENV_LAUNCH_STEP
    }
    HIP_TRY(h, hipGetLastError());
    if (h->reselect_pending) {  // the step after a candidate rebuild: envs whose action was rejected take the parked selection
        hipLaunchKernelGGL(k_apply_reselect, dim3(h->n_envs), dim3(64), 0, h->stream, h->S, h->pend, h->pend_obs, obs_dev);
        HIP_TRY(h, hipGetLastError());
        h->reselect_pending = false;
    }
    h->steps_done += (uint64_t)n_steps;
    if (pos >= 0 && pos == h->timing - 1) {
        HIP_TRY(h, hipEventRecord(h->ev[2 * slot + 1], h->stream));
        h->ev_count += 1;
    }
    return MESHENV_OK;
}

int meshenv_step(MeshEnv *h, const float *actions_dev, float *obs_dev, double *reward_dev, uint8_t *done_dev,
                 uint8_t *complete_dev, float *terminal_obs_dev, int auto_reset)
{
    return launch_step(h, 1, actions_dev, obs_dev, reward_dev, done_dev, complete_dev, terminal_obs_dev, auto_reset);
}

int meshenv_rollout(MeshEnv *h, int n_steps, const float *actions_dev, float *obs_dev, double *reward_dev,
                    uint8_t *done_dev, uint8_t *complete_dev, int auto_reset)
{
    return launch_step(h, n_steps, actions_dev, obs_dev, reward_dev, done_dev, complete_dev, nullptr, auto_reset);
}

__global__ void k_status(DevState S, uint8_t *out)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < S.n_envs) out[e] = (uint8_t)(S.scal[e].status & 3);  // public MESHENV_ST_* bits only
}

int meshenv_get_status(MeshEnv *h, uint8_t *status_dev)
{
    if (!h || !status_dev) return MESHENV_E_ARG;
    MESHENV_ON_DEVICE(h);
    hipLaunchKernelGGL(k_status, dim3((h->n_envs + 255) / 256), dim3(256), 0, h->stream, h->S, status_dev);
    HIP_TRY(h, hipGetLastError());
    return MESHENV_OK;
}

int meshenv_get_state(MeshEnv *h, int env, int32_t *ring_ids_host, double *ring_xy_host, double *cand_key_host,
                      int32_t *cand_stamp_host, int32_t *scalars_host, double *fscalars_host)
{
    if (!h) return MESHENV_E_ARG;
    if (env < 0 || env >= h->n_envs) {
        h->err = "meshenv_get_state: env out of range";
        return MESHENV_E_RANGE;
    }
    MESHENV_ON_DEVICE(h);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    EnvScalars s;
    HIP_TRY(h, hipMemcpy(&s, h->S.scal + env, sizeof(s), hipMemcpyDeviceToHost));
    const size_t off = (size_t)env * (size_t)h->cap;
    const int n0 = h->dom_off_host[s.dom + 1] - h->dom_off_host[s.dom];
    const int n = s.n;
    if (ring_ids_host) {
        HIP_TRY(h, hipMemcpy(ring_ids_host, h->S.ring_id + off, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; i++)  // k-th new vertex -> global id n0 + k (index in boundary.vertices)
            if (ring_ids_host[i] & kNewBit) ring_ids_host[i] = n0 + (ring_ids_host[i] & ~kNewBit);
    }
    if (ring_xy_host) HIP_TRY(h, hipMemcpy(ring_xy_host, h->S.ring_xy + off, sizeof(double2) * n, hipMemcpyDeviceToHost));
    if (cand_stamp_host || cand_key_host) {
        std::vector<int32_t> st((size_t)n);
        std::vector<double> ky((size_t)n);
        HIP_TRY(h, hipMemcpy(st.data(), h->S.ring_stamp + off, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
        HIP_TRY(h, hipMemcpy(ky.data(), h->S.ring_key + off, sizeof(double) * n, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; i++) {
            if (cand_stamp_host) cand_stamp_host[i] = st[i];
            if (cand_key_host) cand_key_host[i] = st[i] == kNotCand ? NAN : ky[i];
        }
    }
    if (scalars_host) {
        scalars_host[0] = s.n;
        scalars_host[1] = s.ref;
        scalars_host[2] = s.n_elem;
        scalars_host[3] = s.failed;
        scalars_host[4] = n0 + s.n_new;
        scalars_host[5] = s.status & 3;
        scalars_host[6] = s.dom;
        scalars_host[7] = n0;
    }
    if (fscalars_host) {
        fscalars_host[0] = s.area;
        fscalars_host[1] = s.bl;
    }
    return MESHENV_OK;
}

// shared body of meshenv_get_elements (which = 0) and meshenv_get_last_episode (which = 1)
static int fetch_elements(MeshEnv *h, const char *fn, int which, int env, int32_t *quads_host, int cap_elems,
                          double *vertex_xy_host, int cap_verts, int32_t *n_elem, int32_t *n_vert, int32_t *flags,
                          int32_t *episodes)
{
    if (!h || !n_elem || !n_vert) return MESHENV_E_ARG;
    if (env < 0 || env >= h->n_envs) {
        h->err = std::string(fn) + ": env out of range";
        return MESHENV_E_RANGE;
    }
    const int cap = h->S.prm.log_cap;
    if (cap <= 0) {
        h->err = std::string(fn) + ": handle was created with log_capacity = 0";
        return MESHENV_E_STATE;
    }
    MESHENV_ON_DEVICE(h);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    EnvScalars s;
    HIP_TRY(h, hipMemcpy(&s, h->S.scal + env, sizeof(s), hipMemcpyDeviceToHost));
    int half = (s.status >> 4) & 1, ne = s.n_elem, nnew = s.n_new;
    if (which) {
        LastEpisode le;
        HIP_TRY(h, hipMemcpy(&le, h->cold.last_ep + env, sizeof(le), hipMemcpyDeviceToHost));
        half ^= 1; ne = le.n_elem; nnew = le.n_new;
        if (flags) *flags = le.flags;
        if (episodes) *episodes = le.episodes;
    }
    const int d = s.dom;
    const int n0 = h->dom_off_host[d + 1] - h->dom_off_host[d];
    ne = ne < cap ? ne : cap;
    nnew = nnew < cap ? nnew : cap;
    if (ne > cap_elems) ne = cap_elems;
    int nv = n0 + nnew;
    if (nv > cap_verts) nv = cap_verts;
    const size_t base = ((size_t)env * 2 + half) * cap;
    if (quads_host && ne > 0) {
        HIP_TRY(h, hipMemcpy(quads_host, h->cold.log_quads + base * 4, sizeof(int32_t) * 4 * (size_t)ne, hipMemcpyDeviceToHost));
        for (int i = 0; i < 4 * ne; i++)
            if (quads_host[i] & kNewBit) quads_host[i] = n0 + (quads_host[i] & ~kNewBit);
    }
    if (vertex_xy_host && nv > 0) {
        const int first = nv < n0 ? nv : n0;
        HIP_TRY(h, hipMemcpy(vertex_xy_host, h->cold.dom_xy + h->dom_off_host[d], sizeof(double2) * (size_t)first, hipMemcpyDeviceToHost));
        if (nv > n0)
            HIP_TRY(h, hipMemcpy(vertex_xy_host + 2 * (size_t)n0, h->cold.log_vxy + base, sizeof(double2) * (size_t)(nv - n0), hipMemcpyDeviceToHost));
    }
    *n_elem = ne;
    *n_vert = nv;
    return MESHENV_OK;
}

int meshenv_get_elements(MeshEnv *h, int env, int32_t *quads_host, int cap_elems, double *vertex_xy_host,
                         int cap_verts, int32_t *n_elem, int32_t *n_vert)
{
    return fetch_elements(h, "meshenv_get_elements", 0, env, quads_host, cap_elems, vertex_xy_host, cap_verts, n_elem,
                          n_vert, nullptr, nullptr);
}

int meshenv_get_last_episode(MeshEnv *h, int env, int32_t *quads_host, int cap_elems, double *vertex_xy_host,
                             int cap_verts, int32_t *n_elem, int32_t *n_vert, int32_t *flags, int32_t *episodes)
{
    return fetch_elements(h, "meshenv_get_last_episode", 1, env, quads_host, cap_elems, vertex_xy_host, cap_verts,
                          n_elem, n_vert, flags, episodes);
}

int meshenv_element_quality(MeshEnv *h, int which, double *elem_dev, double *stats_dev, int32_t *count_dev)
{
    if (!h || (which != 0 && which != 1)) return MESHENV_E_ARG;
    if (h->S.prm.log_cap <= 0) {
        h->err = "meshenv_element_quality: handle was created with log_capacity = 0";
        return MESHENV_E_STATE;
    }
    MESHENV_ON_DEVICE(h);
    hipLaunchKernelGGL(k_element_quality, dim3(h->n_envs), dim3(64), 0, h->stream, h->S, which, elem_dev, stats_dev,
                       count_dev);
    HIP_TRY(h, hipGetLastError());
    return MESHENV_OK;
}

int meshenv_quad_quality(MeshEnv *h, int n, const double *quad_xy_dev, int index, double *out_dev)
{
    if (!h || n < 0 || (n > 0 && (!quad_xy_dev || !out_dev))) return MESHENV_E_ARG;
    if (!(index == 0 || index == 1 || index == 3 || index == 4 || index == 5))
        return fail_arg(h, "meshenv_quad_quality: index must be 0, 1, 3, 4 or 5 (2 and 6 add the boundary term of the extraction "
                           "step: that value is the step's reward)");
    if (n == 0) return MESHENV_OK;
    MESHENV_ON_DEVICE(h);
    hipLaunchKernelGGL(k_quad_quality, dim3((n + 63) / 64), dim3(64), 0, h->stream, n,
                       reinterpret_cast<const double2 *>(quad_xy_dev), index, out_dev);
    HIP_TRY(h, hipGetLastError());
    return MESHENV_OK;
}

int meshenv_counters(MeshEnv *h, uint64_t *out_host)
{
    if (!h || !out_host) return MESHENV_E_ARG;
    MESHENV_ON_DEVICE(h);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    std::vector<EnvCounters> c((size_t)h->n_envs);
    std::vector<EnvScalars> sc((size_t)h->n_envs);
    HIP_TRY(h, hipMemcpy(c.data(), h->S.cnt, sizeof(EnvCounters) * (size_t)h->n_envs, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(sc.data(), h->S.scal, sizeof(EnvScalars) * (size_t)h->n_envs, hipMemcpyDeviceToHost));
    uint64_t a = h->steps_done * (uint64_t)h->n_envs, b = 0, s = 0, sv = 0;
    for (int e = 0; e < h->n_envs; e++) {
        const EnvCounters &k = c[(size_t)e];
        b += k.valid;
        s += k.sum_n + (uint64_t)sc[(size_t)e].n * (h->steps_done - k.last_change);  // the running term of the lazy sum
        sv += k.sum_n_valid;
    }
    out_host[0] = a;
    out_host[1] = b;
    out_host[2] = s;
    out_host[3] = sv;
    return MESHENV_OK;
}

#ifdef MESHENV_SPEC_STATS
// diagnostic build only: k_step_spec event counts / timestamps (reset = 1 clears them)
int meshenv_debug_spec_stats(MeshEnv *h, uint64_t *count_host, uint64_t *time_host, int n_envs, int reset)
{
    if (!h) return MESHENV_E_ARG;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (count_host) HIP_TRY(h, hipMemcpyFromSymbol(count_host, HIP_SYMBOL(g_spec_count), sizeof(uint64_t) * 16));
    if (time_host) HIP_TRY(h, hipMemcpyFromSymbol(time_host, HIP_SYMBOL(g_spec_time), sizeof(uint64_t) * 12 * (size_t)n_envs));
    if (reset) {
        static uint64_t zeros[65536 * 12];
        HIP_TRY(h, hipMemcpyToSymbol(HIP_SYMBOL(g_spec_count), zeros, sizeof(uint64_t) * 16));
        HIP_TRY(h, hipMemcpyToSymbol(HIP_SYMBOL(g_spec_time), zeros, sizeof(uint64_t) * 12 * 65536));
    }
    return MESHENV_OK;
}
#endif

#ifdef MESHENV_STAMPS
// diagnostic build only: raw per-env counter records (see k_step)
int meshenv_debug_raw_counters(MeshEnv *h, uint64_t *out_host)
{
    if (!h || !out_host) return MESHENV_E_ARG;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(out_host, h->S.cnt, sizeof(EnvCounters) * (size_t)h->n_envs, hipMemcpyDeviceToHost));
    return MESHENV_OK;
}
int meshenv_debug_stamps(MeshEnv *h, uint64_t *out_host)
{
    if (!h || !out_host) return MESHENV_E_ARG;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(out_host, h->S.dbg, sizeof(uint64_t) * 16 * (size_t)h->n_envs, hipMemcpyDeviceToHost));
    return MESHENV_OK;
}
#endif

// ---- primitive self-test hook (tests/test_gpu_primitives.py): evaluates the device geometry primitives on
// caller-supplied host arrays so that they can be compared with the CPU oracle's.
__global__ void k_selftest(int what, int n, const double *in, double *out, int libm_exact)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (what == 0) out[i] = round4_py(in[i]);
    else if (what == 1) out[i] = round4_np(in[i]);
    else if (what == 2) out[i] = cw(mkp(in[6 * i], in[6 * i + 1]), mkp(in[6 * i + 2], in[6 * i + 3]), mkp(in[6 * i + 4], in[6 * i + 5]));
    else if (what == 3) out[i] = is_cross(mkp(in[8 * i], in[8 * i + 1]), mkp(in[8 * i + 2], in[8 * i + 3]), mkp(in[8 * i + 4], in[8 * i + 5]),
                                          mkp(in[8 * i + 6], in[8 * i + 7])) ? 1.0 : 0.0;
    else if (what == 4) out[i] = (sin_rounds_to_zero(in[2 * i], in[2 * i + 1]) ? 1.0 : 0.0) + (sin_rounds_to_zero_exact(in[2 * i], in[2 * i + 1]) ? 2.0 : 0.0);
    else if (what == 5) out[i] = (double)round4_npf((float)in[i]);
    else if (what == 6) out[i] = dist(mkp(in[4 * i], in[4 * i + 1]), mkp(in[4 * i + 2], in[4 * i + 3]));
    else if (what == 7) out[i] = (sqrt_pos(in[i]) == sqrt(in[i])) ? 1.0 : 0.0;  // against the compiler's IEEE sqrt
    else if (what == 8) {  // cw_fast against the exact form: 0 equal, 1 guard band raised (and equal after the fallback), 2 MISMATCH
        bool ne;
        const double f = cw_fast(in[2 * i], in[2 * i + 1], ne);
        const double e = cw_exact(in[2 * i], in[2 * i + 1]);
        out[i] = ne ? 1.0 : ((f == e && signbit(f) == signbit(e)) ? 0.0 : 2.0);
    } else if (what == 9 || what == 10) {
        // the front smoother's vertex constructions (csrc/meshenv_smooth.h): 9 doubles = which (0 middle_vertex, 1 side_vertex,
        // 2 indention_vertex), vertex, p1, p2, angle, dist; what 9 -> x, 10 -> y; NaN where the construction is undefined.
        // As in the product path the tan / cos of the angle come from the host's libm: meshenv_selftest replaces the angle
        // by math.tan(math.radians(angle / 2)) (which 0) or math.cos(math.radians(angle)) (which 1, 2) before the upload.
        const double *q = in + 9 * (size_t)i;
        FrontState f;
        f.coord = nullptr; f.adj = nullptr; f.deg = nullptr; f.ringu = nullptr; f.n = 0; f.n0 = 0; f.raised = 0; f.tab = nullptr; f.exact = libm_exact != 0;
        const P2 v = mkp(q[1], q[2]), a = mkp(q[3], q[4]), b = mkp(q[5], q[6]);
        const int which = (int)q[0];
        P2 r;
        if (which == 0) r = middle_vertex(f, v, a, b, q[7]);
        else if (which == 1) r = side_vertex(f, v, a, b, q[7], q[8], false);
        else if (which == 2) r = indention_vertex(f, v, a, b, q[7], q[8], false);
        else {  // 3: Mesh.estimate_4th_vertex(origin, left, right, factor, suggest_dist or < 0 for None)
            const double2 e = estimate_4th_vertex(make_double2(v.x, v.y), make_double2(a.x, a.y), make_double2(b.x, b.y), q[7], q[8] >= 0, q[8]);
            r = mkp(e.x, e.y);
        }
        out[i] = f.raised ? __builtin_nan("") : (what == 9 ? r.x : r.y);
    } else if (what == 11) out[i] = pow2_glibc(in[i]);   // against the host libm's pow(x, 2.0)
    else if (what == 12) out[i] = cw_exact(in[2 * i], in[2 * i + 1]);   // the quantised angle of terms (c, d)
    else if (what == 13) {   // the tie-breaker alone, against the host libm's atan2; NaN outside its domain
        bool ok;
        const double t = atan2_glibc(in[2 * i], in[2 * i + 1], g_atan_cij, ok);
        out[i] = ok ? t : __builtin_nan("");
    } else if (what == 14) out[i] = atan2_cr(in[2 * i], in[2 * i + 1]);
    else if (what == 15) out[i] = sincos_small_nc(in[i]).s;   // against the host libm's sin / cos: within an ulp
    else if (what == 16) out[i] = sincos_small_nc(in[i]).c;
}

int meshenv_selftest(int device, int what, int n, int in_per_item, const double *in_host, double *out_host)
{
    if (n <= 0 || !in_host || !out_host || in_per_item <= 0) return MESHENV_E_ARG;
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return MESHENV_E_HIP;
    double *din = nullptr, *dout = nullptr;
    if (hipMalloc(&din, sizeof(double) * (size_t)n * in_per_item) != hipSuccess) return MESHENV_E_HIP;
    if (hipMalloc(&dout, sizeof(double) * (size_t)n) != hipSuccess) { (void)hipFree(din); return MESHENV_E_HIP; }
    int rc = MESHENV_OK;
    std::vector<double> staged;
    if ((what == 9 || what == 10) && in_per_item == 9) {   // the constructions take the host libm's tan / cos (see k_selftest)
        staged.assign(in_host, in_host + (size_t)n * 9);
        volatile double pi_v = 3.141592653589793;
        const double to_rad = pi_v / 180.0;
        for (int i = 0; i < n; i++) {
            double *q = staged.data() + 9 * (size_t)i;
            const int which = (int)q[0];
            if (which == 0) q[7] = std::tan((q[7] / 2) * to_rad);
            else if (which == 1 || which == 2) q[7] = std::cos(q[7] * to_rad);
        }
        in_host = staged.data();
    }
    if (hipMemcpy(din, in_host, sizeof(double) * (size_t)n * in_per_item, hipMemcpyHostToDevice) != hipSuccess) rc = MESHENV_E_HIP;
    if (rc == MESHENV_OK && upload_atan_state() != hipSuccess) rc = MESHENV_E_HIP;
    if (rc == MESHENV_OK) {
        hipLaunchKernelGGL(k_selftest, dim3((n + 63) / 64), dim3(64), 0, nullptr, what, n, din, dout,
                           (what == 9 || what == 10) ? validate_pow2() : 0);
        if (hipDeviceSynchronize() != hipSuccess) rc = MESHENV_E_HIP;
    }
    if (rc == MESHENV_OK && hipMemcpy(out_host, dout, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) rc = MESHENV_E_HIP;
    (void)hipFree(din);
    (void)hipFree(dout);
    return rc;
}

int meshenv_set_timing(MeshEnv *h, int enable)
{
    if (!h) return MESHENV_E_ARG;
    MESHENV_ON_DEVICE(h);
    if (enable && h->ev.empty()) {
        h->ev.resize(2 * (size_t)MESHENV_TIMING_POOL, nullptr);
        for (hipEvent_t &e : h->ev) HIP_TRY(h, hipEventCreate(&e));
    }
    h->timing = enable > 0 ? enable : 0;
    h->launch_count = 0;
    h->ev_count = 0;
    return MESHENV_OK;
}

int meshenv_kernel_times(MeshEnv *h, float *ms_host, int cap, int32_t *n_out)
{
    if (!h || !ms_host || !n_out || cap < 0) return MESHENV_E_ARG;
    MESHENV_ON_DEVICE(h);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    long long have = h->ev_count < MESHENV_TIMING_POOL ? h->ev_count : MESHENV_TIMING_POOL;
    if (have > cap) have = cap;
    const long long first = h->ev_count - have;
    for (long long k = 0; k < have; k++) {
        const size_t slot = (size_t)((first + k) % MESHENV_TIMING_POOL);
        HIP_TRY(h, hipEventElapsedTime(ms_host + k, h->ev[2 * slot], h->ev[2 * slot + 1]));
        ms_host[k] /= (float)(h->timing > 0 ? h->timing : 1);
    }
    *n_out = (int32_t)have;
    h->ev_count = 0;
    return MESHENV_OK;
}

// ------------------------------------------------------------------------------------------ fused SAC actor
struct MeshActor {
    int device = 0;
    hipStream_t stream = nullptr;
    float *buf = nullptr;  // all weights, one allocation
    size_t nfloat = 0;
    ActorWeights W{};
    bool loaded = false;
    std::string err;
};

int meshenv_actor_create(int device, void *stream, MeshActor **out)
{
    if (!out) return MESHENV_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        g_create_error = "meshenv_actor_create: no such HIP device";
        return MESHENV_E_HIP;
    }
    MeshActor *a = new MeshActor();
    a->device = device;
    a->stream = (hipStream_t)stream;
    a->nfloat = (size_t)kActInPad * kActHid + kActHid + 2 * ((size_t)kActHid * kActHid + kActHid) + (size_t)kActHid * 16 + 16;
    DeviceGuard guard(device);
    if (guard.err != hipSuccess || hipMalloc((void **)&a->buf, a->nfloat * sizeof(float)) != hipSuccess) {
        g_create_error = "meshenv_actor_create: hipMalloc failed";
        delete a;
        return MESHENV_E_HIP;
    }
    *out = a;
    return MESHENV_OK;
}

void meshenv_actor_destroy(MeshActor *a)
{
    if (!a) return;
    DeviceGuard guard(a->device);
    (void)hipStreamSynchronize(a->stream);
    if (a->buf) (void)hipFree(a->buf);
    delete a;
}

int meshenv_actor_set_stream(MeshActor *a, void *stream)
{
    if (!a) return MESHENV_E_ARG;
    a->stream = (hipStream_t)stream;
    return MESHENV_OK;
}

// weights in torch.nn.Linear layout ([out][in], row-major), host pointers
int meshenv_actor_load(MeshActor *a, const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                       const float *b3, const float *w_mu, const float *b_mu, const float *w_log_std,
                       const float *b_log_std, const float *low, const float *high)
{
    if (!a || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !w_mu || !b_mu || !w_log_std || !b_log_std || !low || !high)
        return MESHENV_E_ARG;
    std::vector<float> h;
    // torch [out][in] -> the per-lane MFMA B-operand order of meshenv_actor.h: [16-neuron tile = 2 wv + tile][K/16][lane][4]
    auto packed = [&](const float *w, int out, int in, int k_pad, int waves) {
        const size_t off = h.size();
        const int groups = k_pad / 16;
        h.resize(off + (size_t)waves * 2 * groups * 64 * 4, 0.0f);
        for (int wv = 0; wv < waves; wv++)
            for (int tile = 0; tile < 2; tile++)
                for (int g = 0; g < groups; g++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int j = 0; j < 4; j++) {
                            const int k = 4 * (4 * g + j) + (lane >> 4), n = 32 * wv + 16 * tile + (lane & 15);
                            if (k < in && n < out)
                                h[off + ((((size_t)wv * 2 + tile) * groups + g) * 64 + lane) * 4 + j] = w[(size_t)n * in + k];
                        }
        return off;
    };
    auto plain = [&](const float *b, int n, int pad) {
        const size_t off = h.size();
        h.resize(off + pad, 0.0f);
        for (int i = 0; i < n; i++) h[off + i] = b[i];
        return off;
    };
    const size_t o_w1 = packed(w1, kActHid, kActIn, kActInPad, 4), o_b1 = plain(b1, kActHid, kActHid);
    const size_t o_w2 = packed(w2, kActHid, kActHid, kActHid, 4), o_b2 = plain(b2, kActHid, kActHid);
    const size_t o_w3 = packed(w3, kActHid, kActHid, kActHid, 4), o_b3 = plain(b3, kActHid, kActHid);
    // heads: one 16-wide tile, columns mu0..2, log_std0..2, zeros: [8][64][4]
    std::vector<float> wh((size_t)16 * kActHid, 0.0f);  // [n 16][k 128]
    for (int o = 0; o < 3; o++)
        for (int k = 0; k < kActHid; k++) {
            wh[(size_t)o * kActHid + k] = w_mu[(size_t)o * kActHid + k];
            wh[(size_t)(3 + o) * kActHid + k] = w_log_std[(size_t)o * kActHid + k];
        }
    const size_t o_wh = h.size();
    h.resize(o_wh + (size_t)8 * 64 * 4, 0.0f);
    for (int g = 0; g < 8; g++)
        for (int lane = 0; lane < 64; lane++)
            for (int j = 0; j < 4; j++)
                h[o_wh + ((size_t)g * 64 + lane) * 4 + j] = wh[(size_t)(lane & 15) * kActHid + 4 * (4 * g + j) + (lane >> 4)];
    const size_t o_bh = h.size();
    h.resize(o_bh + 16, 0.0f);
    for (int o = 0; o < 3; o++) { h[o_bh + o] = b_mu[o]; h[o_bh + 3 + o] = b_log_std[o]; }
    if (h.size() > a->nfloat) {
        a->err = "meshenv_actor_load: internal size mismatch";
        return MESHENV_E_STATE;
    }
    DeviceGuard guard(a->device);
    if (guard.err != hipSuccess ||
        hipMemcpy(a->buf, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        a->err = "meshenv_actor_load: upload failed";
        return MESHENV_E_HIP;
    }
    a->W.w1p = a->buf + o_w1; a->W.b1 = a->buf + o_b1;
    a->W.w2p = a->buf + o_w2; a->W.b2 = a->buf + o_b2;
    a->W.w3p = a->buf + o_w3; a->W.b3 = a->buf + o_b3;
    a->W.whp = a->buf + o_wh; a->W.bh = a->buf + o_bh;
    for (int i = 0; i < 3; i++) { a->W.low[i] = low[i]; a->W.high[i] = high[i]; }
    a->loaded = true;
    return MESHENV_OK;
}

static int actor_launch(MeshActor *a, const char *fn, int n, const float *obs_dev, const float *noise_dev,
                        float *actions_dev, int sample, uint64_t seed, uint64_t counter, float *eps_out_dev)
{
    if (!a || n <= 0 || !obs_dev || !actions_dev) return MESHENV_E_ARG;
    if (!a->loaded) {
        a->err = std::string(fn) + ": no weights loaded";
        return MESHENV_E_STATE;
    }
    DeviceGuard guard(a->device);
    if (guard.err != hipSuccess) {
        a->err = std::string(fn) + ": hipSetDevice failed";
        return MESHENV_E_HIP;
    }
    hipLaunchKernelGGL(k_actor_forward, dim3((n + kActEnvs - 1) / kActEnvs), dim3(64 * kActWaves), 0, a->stream, a->W, n, obs_dev, noise_dev,
                       actions_dev, sample, seed, counter, eps_out_dev);
    if (hipGetLastError() != hipSuccess) {
        a->err = std::string(fn) + ": launch failed";
        return MESHENV_E_HIP;
    }
    return MESHENV_OK;
}

int meshenv_actor_forward(MeshActor *a, int n, const float *obs_dev, const float *noise_dev, float *actions_dev)
{
    return actor_launch(a, "meshenv_actor_forward", n, obs_dev, noise_dev, actions_dev, 0, 0, 0, nullptr);
}

int meshenv_actor_sample(MeshActor *a, int n, const float *obs_dev, uint64_t seed, uint64_t counter, float *actions_dev,
                         float *eps_out_dev)
{
    return actor_launch(a, "meshenv_actor_sample", n, obs_dev, nullptr, actions_dev, 1, seed, counter, eps_out_dev);
}

int meshenv_step_actor(MeshEnv *h, MeshActor *a, const float *actions_dev, float *obs_dev, double *reward_dev, uint8_t *done_dev,
                       uint8_t *complete_dev, float *terminal_obs_dev, int auto_reset, int sample, uint64_t seed, uint64_t counter,
                       float *actions_next_dev, float *eps_out_dev)
{
    if (!h || !a) return MESHENV_E_ARG;
    if (!actions_dev || !obs_dev || !reward_dev || !done_dev || !complete_dev || !actions_next_dev || actions_next_dev == actions_dev)
        return fail_arg(h, "meshenv_step_actor: null device pointer, or actions_next aliases actions");
    if (!a->loaded) {
        h->err = "meshenv_step_actor: the actor has no weights loaded";
        return MESHENV_E_STATE;
    }
    if (a->device != h->device || a->stream != h->stream) {   // the two halves are ordered by the stream alone
        h->err = "meshenv_step_actor: env and actor must be on the same device and stream (meshenv_set_stream / meshenv_actor_set_stream)";
        return MESHENV_E_STATE;
    }
    const bool fusable = h->group == 16 && !h->spec && h->default_params && !h->front_moved && !h->env_lds &&
                         h->timing == 0 && !h->reselect_pending && group_actor_lds_bytes(h->cap) <= 160 * 1024;
    if (!fusable) {   // same results by two launches (other batch sizes / ring lengths, timing armed, the
                      // step after meshenv_smooth whose parked re-selection changes the observation the policy reads)
        const int rc = launch_step(h, 1, actions_dev, obs_dev, reward_dev, done_dev, complete_dev, terminal_obs_dev, auto_reset);
        if (rc != MESHENV_OK) return rc;
        const int ra = actor_launch(a, "meshenv_step_actor", h->n_envs, obs_dev, nullptr, actions_next_dev, sample, seed, counter,
                                    eps_out_dev);
        if (ra != MESHENV_OK) h->err = a->err;
        return ra;
    }
    MESHENV_ON_DEVICE(h);
    if (!h->fused_ready) {
        HIP_TRY(h, hipFuncSetAttribute((const void *)k_step_group_actor<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_TRY(h, hipFuncSetAttribute((const void *)k_step_group_actor<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        h->fused_ready = true;
    }
    GroupActorArgs GA;
    GA.g.S = h->S;
    GA.g.outs.obs_out = obs_dev; GA.g.outs.reward = reward_dev; GA.g.outs.done = done_dev; GA.g.outs.complete = complete_dev;
    GA.g.outs.term_obs = terminal_obs_dev;
    GA.g.actions = actions_dev;
    GA.g.step0 = (unsigned long long)h->steps_done;
    GA.g.cap = h->cap;
    GA.g.env_lds = nullptr; GA.g.ho_off = 0; GA.g.pad = 0;
    GA.g.auto_reset = auto_reset;
    GA.W = a->W;
    GA.actions_next = actions_next_dev;
    GA.eps_out = eps_out_dev;
    GA.seed = seed; GA.counter = counter;
    GA.sample = sample ? 1 : 0; GA.pad = 0;
    if (h->cap <= 64) hipLaunchKernelGGL((k_step_group_actor<true, true>), dim3((h->n_envs + 15) / 16), dim3(64 * 16), group_actor_lds_bytes(h->cap), h->stream, GA);
    else hipLaunchKernelGGL((k_step_group_actor<true>), dim3((h->n_envs + 15) / 16), dim3(64 * 16), group_actor_lds_bytes(h->cap), h->stream, GA);
    HIP_TRY(h, hipGetLastError());
    h->steps_done += 1;
    return MESHENV_OK;
}

int meshenv_extract_samples(MeshEnv *h, int which, const uint8_t *mask_dev, int n_neighbor, int n_radius, double radius, int index,
                            double quality_threshold, int64_t *count_dev, uint8_t *status_dev, const int64_t *offsets_dev,
                            double *samples_dev, double *outputs_dev, double *types_dev)
{
    if (!h || (which != 0 && which != 1) || !count_dev || !status_dev) return MESHENV_E_ARG;
    if (n_neighbor < 1 || n_neighbor > kSampMaxNeighbor || n_radius < 1 || n_radius > kSampMaxRadius || !(radius > 0) ||
        (index != 1 && index != 5))
        return fail_arg(h, "meshenv_extract_samples: n_neighbor in 1..3, n_radius in 1..4, radius > 0 and index 1 or 5 (the values of the reference's callers) are supported");
    if (offsets_dev && (!samples_dev || !outputs_dev || !types_dev))
        return fail_arg(h, "meshenv_extract_samples: offsets_dev given without the three output arrays");
    const int log_cap = h->S.prm.log_cap;
    if (log_cap <= 0) {
        h->err = "meshenv_extract_samples: handle was created with log_capacity = 0 (the mesh graph is rebuilt from the element log)";
        return MESHENV_E_STATE;
    }
    const size_t lds = samples_lds_bytes(h->cap, log_cap);
    if (lds > 160 * 1024 || h->cap + log_cap > 65535)
        return fail_arg(h, "meshenv_extract_samples: ring stride + log_capacity too large for the kernel's LDS (49 B per vertex)");
    MESHENV_ON_DEVICE(h);
    int rc = ensure_libm_tables(h);
    if (rc != MESHENV_OK) return rc;
    if (!h->samples_ready) {
        HIP_TRY(h, hipFuncSetAttribute((const void *)k_extract_samples<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_TRY(h, hipFuncSetAttribute((const void *)k_extract_samples<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        h->samples_ready = true;
    }
    SampleParams P;
    P.n_neighbor = n_neighbor; P.n_radius = n_radius; P.index = index; P.radius = radius; P.quality_threshold = quality_threshold;
    const dim3 grid(h->n_envs), block(64);
    if (!offsets_dev)
        hipLaunchKernelGGL(k_extract_samples<false>, grid, block, lds, h->stream, h->S, h->cap, which, mask_dev, P, libm_ref(h),
                           (long long *)count_dev, status_dev, (const long long *)nullptr, (double *)nullptr, (double *)nullptr, (double *)nullptr);
    else
        hipLaunchKernelGGL(k_extract_samples<true>, grid, block, lds, h->stream, h->S, h->cap, which, mask_dev, P, libm_ref(h),
                           (long long *)count_dev, status_dev, (const long long *)offsets_dev, samples_dev, outputs_dev, types_dev);
    HIP_TRY(h, hipGetLastError());
    return MESHENV_OK;
}

#ifdef MESHENV_DEV
static unsigned long long *g_tsteps_dbg = nullptr;
#endif

int meshenv_step_actor_multi(MeshEnv *h, MeshActor *a, int T, float *actions_dev, float *obs_dev, double *reward_dev, uint8_t *done_dev,
                         uint8_t *complete_dev, float *terminal_obs_dev, int auto_reset, int sample, uint64_t seed, uint64_t counter,
                         float *eps_out_dev)
{
    if (!h || !a) return MESHENV_E_ARG;
    if (T <= 0 || !actions_dev || !obs_dev || !reward_dev || !done_dev || !complete_dev)
        return fail_arg(h, "meshenv_step_actor_multi: T > 0 and non-null device pointers are required");
    if (!a->loaded) {
        h->err = "meshenv_step_actor_multi: the actor has no weights loaded";
        return MESHENV_E_STATE;
    }
    if (a->device != h->device || a->stream != h->stream) {
        h->err = "meshenv_step_actor_multi: env and actor must be on the same device and stream (meshenv_set_stream / meshenv_actor_set_stream)";
        return MESHENV_E_STATE;
    }
    const size_t n = (size_t)h->n_envs;
    // One launch for all T steps (k_step_group_actor_T) measured SLOWER than T fused single-step launches (DESIGN.md section 5,
    // round 3: 95 spilled VGPRs, 384 B of scratch per lane): that kernel is compiled in the -DMESHENV_DEV build only, where
    // tools/tsteps_timeline.py and tests/test_gpu_variants.py drive it; the shipped library always takes the step-by-step form.
#ifdef MESHENV_DEV
    const bool fusable = h->group == 16 && !h->spec && h->default_params && !h->front_moved && !h->env_lds && h->timing == 0 && !h->reselect_pending &&
                         group_actor_lds_bytes(h->cap) <= 160 * 1024 && !h->S.msg;
#else
    const bool fusable = false;
#endif
    if (!fusable || T == 1) {   // the same results step by step
        for (int t = 0; t < T; t++) {
            const int rc = meshenv_step_actor(h, a, actions_dev + (size_t)t * n * 3, obs_dev + (size_t)t * n * kObsDim, reward_dev + (size_t)t * n,
                                              done_dev + (size_t)t * n, complete_dev + (size_t)t * n,
                                              terminal_obs_dev ? terminal_obs_dev + (size_t)t * n * kObsDim : nullptr, auto_reset, sample, seed,
                                              counter + (uint64_t)t, actions_dev + (size_t)(t + 1) * n * 3,
                                              eps_out_dev ? eps_out_dev + (size_t)t * n * 3 : nullptr);
            if (rc != MESHENV_OK) return rc;
        }
        return MESHENV_OK;
    }
#ifdef MESHENV_DEV
    MESHENV_ON_DEVICE(h);
    if (!h->fused_T_ready) {
        HIP_TRY(h, hipFuncSetAttribute((const void *)k_step_group_actor_T<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        h->fused_T_ready = true;
    }
    GroupActorArgsT GA;
    GA.g.S = h->S;
    GA.g.outs.obs_out = obs_dev; GA.g.outs.reward = reward_dev; GA.g.outs.done = done_dev; GA.g.outs.complete = complete_dev;
    GA.g.outs.term_obs = terminal_obs_dev;
    GA.g.actions = actions_dev;
    GA.g.step0 = (unsigned long long)h->steps_done;
    GA.g.cap = h->cap;
    GA.g.env_lds = nullptr; GA.g.ho_off = 0; GA.g.pad = 0;
    GA.g.auto_reset = auto_reset;
    GA.W = a->W;
    GA.eps_out = eps_out_dev;
    GA.seed = seed; GA.counter = counter;
    GA.sample = sample ? 1 : 0;
    GA.T = T;
    GA.dbg = nullptr;
    GA.dbg = g_tsteps_dbg;   // nullptr unless tools/tsteps_timeline.py set a stamp buffer (meshenv_dev_set_tsteps_dbg)
    hipLaunchKernelGGL((k_step_group_actor_T<true>), dim3((h->n_envs + 15) / 16), dim3(64 * 16), group_actor_lds_bytes(h->cap), h->stream, GA);
    HIP_TRY(h, hipGetLastError());
    h->steps_done += (uint64_t)T;
#endif
    return MESHENV_OK;
}

#ifdef MESHENV_DEV
// Dev build only (tools/tsteps_timeline.py): a device buffer of at least grid * T * 4 u64 words that the next
// meshenv_step_actor_multi launches stamp; nullptr switches the stamps off.  Not part of include/meshenv.h.
int meshenv_dev_set_tsteps_dbg(unsigned long long *buf_dev)
{
    g_tsteps_dbg = buf_dev;
    return MESHENV_OK;
}
#endif

}  // extern "C"
