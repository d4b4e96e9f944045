// gsss_capi.hip -- the extern "C" surface declared in include/gsss.h, plus the layout and
// initial-state kernels.  No torch types, no exceptions across the boundary.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "gsss_fast.h"
#include "gsss_launch.h"
#include "gsss_mh.h"

namespace gsss {

// ------------------------------------------------------------------------------------------
// error text (thread local)
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static thread_local LaunchInfo g_last_launch = {0, 0, 0.0};
LaunchInfo &last_launch() { return g_last_launch; }

int64_t resident_workgroups(const void *kern, size_t lds_bytes, int *per_cu_out)
{
    struct Entry {
        const void *kern;
        size_t lds;
        int dev, per_cu, cus;
    };
    static std::mutex mu;
    static std::vector<Entry> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    // GSSS_RESIDENT_PER_CU (tests): plan the launch as if the chip held this many workgroups of the kernel per CU -- the sliced
    // shapes of another box reproduced on this one.  Safe for any value: a slice waits only for workgroups that drew their ticket
    // before it (SliceSched), whatever the plan assumed to be resident.  Read per call, never cached.
    const char *env = getenv("GSSS_RESIDENT_PER_CU");
    const int forced = env ? atoi(env) : 0;
    int per_cu = 0, cus = 0;
    {
        std::lock_guard<std::mutex> lock(mu);
        for (const Entry &e : cache)
            if (e.kern == kern && e.lds == lds_bytes && e.dev == dev) {
                per_cu = e.per_cu;
                cus = e.cus;
                break;
            }
    }
    if (per_cu < 1) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kBlock, lds_bytes) != hipSuccess || per_cu < 1 ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) {
            (void)hipGetLastError();
            return 0;
        }
        std::lock_guard<std::mutex> lock(mu);
        cache.push_back(Entry{kern, lds_bytes, dev, per_cu, cus});
    }
    if (forced > 0) per_cu = forced;
    if (per_cu_out) *per_cu_out = per_cu;
    return (int64_t)per_cu * cus;
}

void slice_fallback_note(const char *why)
{
    if (getenv("GSSS_DEBUG_OCCUPANCY")) fprintf(stderr, "gsss: %s\n", why);
}

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ------------------------------------------------------------------------------------------
// vector layout table / selection
// ------------------------------------------------------------------------------------------
static const VecInfo kVecs[] = {
#define GSSS_VEC_ROW(ID, V, NAME) {ID, V::L, V::N, V::DPAD, V::kExactDim, NAME},
    GSSS_VEC_LIST(GSSS_VEC_ROW)
#undef GSSS_VEC_ROW
};

const VecInfo *vec_table(int *count)
{
    *count = (int)(sizeof(kVecs) / sizeof(kVecs[0]));
    return kVecs;
}

int select_vec(int d, int variant)
{
    int n;
    const VecInfo *v = vec_table(&n);
    if (variant != 0) {
        for (int i = 0; i < n; ++i)
            if (v[i].id == variant) {
                if (v[i].exact_dim ? v[i].dpad == d : d <= v[i].dpad) return variant;
                set_error("kernel variant %s does not cover d=%d", v[i].name, d);
                return GSSS_E_UNSUPPORTED;
            }
        set_error("unknown kernel variant %d", variant);
        return GSSS_E_INVALID;
    }
    for (int i = 0; i < n; ++i)  // lane layouts first (exact d), then the smallest cooperative one
        if (v[i].exact_dim && v[i].dpad == d) return v[i].id;
    for (int i = 0; i < n; ++i)
        if (!v[i].exact_dim && d <= v[i].dpad) return v[i].id;
    set_error("no kernel variant covers d=%d (max %d)", d, v[n - 1].dpad);
    return GSSS_E_UNSUPPORTED;
}

// The layout an exact-mode launch of this target runs in: select_vec's -- unless the target's rows (component means, knots, a
// dense A) do not fit a workgroup's LDS in it: then the smallest sixty-four-lane layout, whose kernels read such rows from global
// memory (rows_fit_lds, gsss_device.h).  A layout forced by the caller stays as asked.
static int select_vec_for(const gsss::TargetBlock &tb, int variant)
{
    const int vec = select_vec(tb.d, variant);
    if (vec < 0 || variant != 0) return vec;
    int n;
    const VecInfo *v = vec_table(&n);
    size_t dpad = 0;
    for (int i = 0; i < n; ++i)
        if (v[i].id == vec) dpad = (size_t)v[i].dpad;
    if (dpad >= 256) return vec;
    size_t rows = 0;
    switch (tb.kind) {
    case GSSS_VMF_MIXTURE: rows = (size_t)tb.k * dpad + (size_t)tb.k; break;
    case GSSS_BINGHAM: rows = (size_t)(tb.d + 1) * dpad; break;
    case GSSS_CURVE_VMF: rows = (size_t)tb.k * dpad + 4 * (size_t)(tb.k - 1); break;
    default: return vec;
    }
    if (rows * sizeof(double) <= gsss::kRowsLdsBytes) return vec;
    for (int i = 0; i < n; ++i)
        if (!v[i].exact_dim && v[i].dpad >= 256 && tb.d <= v[i].dpad) return v[i].id;
    return vec;
}

// ------------------------------------------------------------------------------------------
// device guard: HIP's current device is per host thread; leave it as we found it
// ------------------------------------------------------------------------------------------
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(device) == hipSuccess;
        if (!ok) set_error("hipSetDevice(%d) failed", device);
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// ------------------------------------------------------------------------------------------
// layout kernels: out[c][r] = in[r][c] for a row-major [R][C] matrix of doubles, LDS tiled so
// that both the read and the write are 64-lane coalesced (HBM-bound; 16 B of traffic per element)
// ------------------------------------------------------------------------------------------
constexpr int kTile = 64;

__global__ void __launch_bounds__(kBlock) transpose_kernel(const double *__restrict__ in, double *__restrict__ out,
                                                           int64_t R, int64_t C)
{
    __shared__ double tile[kTile][kTile + 1];
    const int64_t gx = (C + kTile - 1) / kTile;
    const int64_t r0 = ((int64_t)blockIdx.x / gx) * kTile, c0 = ((int64_t)blockIdx.x % gx) * kTile;
    const int tx = threadIdx.x % kTile, ty = threadIdx.x / kTile;  // 64 x 4
    for (int i = ty; i < kTile; i += kBlock / kTile) {
        const int64_t r = r0 + i, c = c0 + tx;
        if (r < R && c < C) tile[i][tx] = in[r * C + c];
    }
    __syncthreads();
    for (int i = ty; i < kTile; i += kBlock / kTile) {
        const int64_t c = c0 + i, r = r0 + tx;
        if (r < R && c < C) out[c * R + r] = tile[tx][i];
    }
}

// 16-byte variant of the tile transpose for even R, C and 16-byte aligned buffers: every lane moves
// a double2 on both sides (1 KiB per wave-instruction), the tile is transposed on its way INTO LDS.
__global__ void __launch_bounds__(kBlock) transpose_kernel_x2(const double *__restrict__ in, double *__restrict__ out,
                                                              int64_t R, int64_t C)
{
    __shared__ double tileT[kTile][kTile + 2];  // [column][row], row stride 66 doubles keeps 16-B alignment
    const int64_t gx = (C + kTile - 1) / kTile;
    const int64_t r0 = ((int64_t)blockIdx.x / gx) * kTile, c0 = ((int64_t)blockIdx.x % gx) * kTile;
    const int tx = threadIdx.x % 32, ty = threadIdx.x / 32;  // 32 x 8
    for (int i = ty; i < kTile; i += 8) {
        const int64_t r = r0 + i, c = c0 + 2 * tx;
        if (r < R && c < C) {  // C even: c + 1 < C too
            const double2 v = *reinterpret_cast<const double2 *>(in + r * C + c);
            tileT[2 * tx][i] = v.x;
            tileT[2 * tx + 1][i] = v.y;
        }
    }
    __syncthreads();
    for (int j = ty; j < kTile; j += 8) {
        const int64_t c = c0 + j, r = r0 + 2 * tx;
        if (c < C && r < R) {
            const double2 v = *reinterpret_cast<const double2 *>(&tileT[j][2 * tx]);
            *reinterpret_cast<double2 *>(out + c * R + r) = v;
        }
    }
}

// Skinny matrices (one side <= 32, e.g. [n][3] <-> [3][n]): a 64x64 tile would be mostly empty.
// A workgroup moves 256 indices of the long side x all S of the short side through LDS so that
// the side that is contiguous in memory is read / written as one linear, fully coalesced run.
constexpr int kSkinny = 32;

template <bool SHORT_ROWS>
__global__ void __launch_bounds__(kBlock) transpose_skinny_kernel(const double *__restrict__ in,
                                                                  double *__restrict__ out, int64_t R, int64_t C)
{
    __shared__ double buf[kBlock * (kSkinny + 1)];
    const int S = (int)(SHORT_ROWS ? R : C);           // short side
    const int64_t Lg = SHORT_ROWS ? C : R;              // long side
    const int SP = S | 1;                               // odd LDS stride: conflict-free
    const int64_t l0 = (int64_t)blockIdx.x * kBlock;
    const int nl = (int)((Lg - l0) < kBlock ? (Lg - l0) : kBlock);
    const int tid = threadIdx.x;
    if (SHORT_ROWS) {
        // in[r][l] coalesced along l; out[l][r]: one contiguous run of nl*S doubles
        if (tid < nl)
            for (int r = 0; r < S; ++r) buf[tid * SP + r] = in[(int64_t)r * C + l0 + tid];
        __syncthreads();
        double *dst = out + l0 * S;
        for (int k = tid; k < nl * S; k += kBlock) dst[k] = buf[(k / S) * SP + (k % S)];
    } else {
        // in[l][c]: one contiguous run of nl*S doubles; out[c][l] coalesced along l
        const double *src = in + l0 * S;
        for (int k = tid; k < nl * S; k += kBlock) buf[(k / S) * SP + (k % S)] = src[k];
        __syncthreads();
        if (tid < nl)
            for (int c = 0; c < S; ++c) out[(int64_t)c * R + l0 + tid] = buf[tid * SP + c];
    }
}

static int transpose(const double *in, double *out, int64_t R, int64_t C, int device, hipStream_t st)
{
    if (R <= 0 || C <= 0) return GSSS_OK;
    if (!in || !out) {
        set_error("null buffer");
        return GSSS_E_INVALID;
    }
    DeviceGuard guard(device);
    if (!guard.ok) return GSSS_E_HIP;
    if (R <= kSkinny || C <= kSkinny) {
        const bool short_rows = R <= C;
        const int64_t longside = short_rows ? C : R;
        const int64_t grid = (longside + kBlock - 1) / kBlock;
        if (grid > 0x7FFFFFFFll) {
            set_error("transpose: %lld x %lld exceeds the grid", (long long)R, (long long)C);
            return GSSS_E_UNSUPPORTED;
        }
        if (short_rows)
            hipLaunchKernelGGL(transpose_skinny_kernel<true>, dim3((unsigned)grid), dim3(kBlock), 0, st, in, out, R, C);
        else
            hipLaunchKernelGGL(transpose_skinny_kernel<false>, dim3((unsigned)grid), dim3(kBlock), 0, st, in, out, R, C);
        GSSS_HIP_TRY(hipGetLastError());
        return GSSS_OK;
    }
    const int64_t gx = (C + kTile - 1) / kTile, gy = (R + kTile - 1) / kTile;
    if (gx * gy > 0x7FFFFFFFll) {
        set_error("transpose: %lld x %lld exceeds the grid", (long long)R, (long long)C);
        return GSSS_E_UNSUPPORTED;
    }
    const bool wide = (R % 2 == 0) && (C % 2 == 0) && (reinterpret_cast<uintptr_t>(in) % 16 == 0) &&
                      (reinterpret_cast<uintptr_t>(out) % 16 == 0);
    if (wide)
        hipLaunchKernelGGL(transpose_kernel_x2, dim3((unsigned)(gx * gy)), dim3(kBlock), 0, st, in, out, R, C);
    else
        hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)(gx * gy)), dim3(kBlock), 0, st, in, out, R, C);
    GSSS_HIP_TRY(hipGetLastError());
    return GSSS_OK;
}

// ------------------------------------------------------------------------------------------
// sphere.sample_sphere twin: normals from the reserved step id, radially projected
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) sample_sphere_kernel(uint64_t seed, uint64_t chain_offset, int64_t n, int d,
                                                               double *__restrict__ out)
{
    const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (c >= n) return;
    RunBlock rb{};
    rb.seed = seed;
    rb.chain_offset = chain_offset;
    PhiloxDraws<LaneVec<2>> dr;
    dr.init(rb, c, d);
    dr.begin_step(kInitStep);
    double ss = 0.0;
    for (int q = 0; 4 * q < d; ++q) {
        uint32_t w[4];
        double zz[4];
        dr.words(1u + (uint32_t)q, w);
        box_muller32(w[0], w[1], zz[0], zz[1]);
        box_muller32(w[2], w[3], zz[2], zz[3]);
        for (int i = 0; i < 4 && 4 * q + i < d; ++i) {
            out[(size_t)(4 * q + i) * n + c] = zz[i];
            ss = fma(zz[i], zz[i], ss);
        }
    }
    const double nrm = sqrt(ss) + 1e-100;  // sphere.py:14
    for (int j = 0; j < d; ++j) out[(size_t)j * n + c] = out[(size_t)j * n + c] / nrm;
}


// ------------------------------------------------------------------------------------------
// The S^2 tangent draw of the Philox stream, exposed for verification (gsss_tangent_s2): the very expressions the set-up of a
// step evaluates (screened_kernel / fast_kernel / wave_kernel: x.x, inv_norm, x / |x|, PhiloxDraws::tangent -> tangent3) for
// given states and angle words.  out[c][0..2] = n = x / |x|, [3..5] = b1 (the tangent at angle 0), [6..8] = b2 (at a quarter
// turn), [9..11] = the tangent for word w[c].  tab: the table-driven sincos of the throughput kernels, else the exact kernels'.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) tangent_s2_kernel(const double *__restrict__ x, const uint32_t *__restrict__ w, int64_t n,
                                                            int tab_driven, double *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) double lds[kTabLds + 2];
    const fm::Tables tab = stage_tables(lds);
    __syncthreads();
    const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (c >= n) return;
    using V = LaneVec<3>;
    const double xs[3] = {x[3 * c], x[3 * c + 1], x[3 * c + 2]};
    const double xx = vdot<V>(xs, xs);
    double nrm[3];
    if (tab_driven) {  // gsss_screen.h / gsss_fast.h: x * rsqrt(x.x)
        const double rnx = inv_norm(xx);
        for (int j = 0; j < 3; ++j) nrm[j] = xs[j] * rnx;
    } else {           // run_kernel (gsss_device.h): x / (|x| + 1e-100), sphere.py:14
        const double nx = sqrt(xx) + 1e-100;
        for (int j = 0; j < 3; ++j) nrm[j] = xs[j] / nx;
    }
    double u[3], b1[3], b2[3];
    if (tab_driven) {
        PhiloxDraws<V, true> dr;
        dr.tab = tab;
        dr.d = 3;
        dr.tangent(nrm, u, 0, w[c]);
    } else {
        PhiloxDraws<V, false> dr;
        dr.d = 3;
        dr.tangent(nrm, u, 0, w[c]);
    }
    tangent3(nrm[0], nrm[1], nrm[2], 0.0, 1.0, b1[0], b1[1], b1[2]);
    tangent3(nrm[0], nrm[1], nrm[2], 1.0, 0.0, b2[0], b2[1], b2[2]);
    double *o = out + 12 * c;
    for (int j = 0; j < 3; ++j) {
        o[j] = nrm[j];
        o[3 + j] = b1[j];
        o[6 + j] = b2[j];
        o[9 + j] = u[j];
    }
}

}  // namespace gsss

namespace gsss {
int launch_cpd_run(int variant, int draws, const TargetBlock &tb, const RunBlock &rb, hipStream_t st);
int launch_cpd_mh(int variant, int draws, int sampler, const TargetBlock &tb, const RunBlock &rb, const MhBlock &mb, hipStream_t st);
int launch_cpd_logprob(int variant, const TargetBlock &tb, const double *x, int64_t n, double *out, bool grad, hipStream_t st);
}  // namespace gsss

static int fast_dispatch(const gsss::TargetBlock &tb, const gsss::RunBlock &rb, bool replay, gsss::FastProbe *probe,
                         hipStream_t st)
{
    switch (tb.kind) {
    case GSSS_VMF_MIXTURE: return gsss::launch_fast_vmf(tb, rb, replay, probe, st);
    case GSSS_BINGHAM: return gsss::launch_fast_bingham(tb, rb, replay, probe, st);
    case GSSS_CURVE_VMF: return gsss::launch_fast_curve(tb, rb, replay, probe, st);
    }
    if (!probe) gsss::set_error("fast mode is not built for target kind %d", tb.kind);
    return GSSS_E_UNSUPPORTED;
}

// ==========================================================================================
// extern "C"
// ==========================================================================================
using namespace gsss;

struct gsss_target {
    int device;
    TargetBlock tb;
    double *blob_dev;
    size_t blob_doubles;
    int cpd_variant;  // GSSS_CPD: which registration kernel (neighbour-list size, uniform source weights)
};

extern "C" {

int gsss_abi_version(void) { return GSSS_ABI_VERSION; }

const char *gsss_last_error(void) { return g_err; }

int gsss_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int gsss_target_create(const gsss_target_desc *desc, int device, gsss_target **out)
{
    if (!desc || !out) {
        set_error("null argument");
        return GSSS_E_INVALID;
    }
    *out = nullptr;
    const int d = desc->d, k = desc->k;
    if (d < 2) {
        set_error("d must be >= 2 (got %d)", d);
        return GSSS_E_INVALID;
    }
    if (select_vec(d, 0) < 0) return GSSS_E_UNSUPPORTED;  // before any parameter array is touched
    std::vector<double> blob;
    bool bingham_diagonal = false;
    double scale = 0.0;
    int cpd_variant = 0;
    switch (desc->kind) {
    case GSSS_VMF_MIXTURE:
        if (k < 1 || !desc->mu || !desc->logc) {
            set_error("vMF mixture needs k >= 1, mu and logc");
            return GSSS_E_INVALID;
        }
        blob.assign(desc->mu, desc->mu + (size_t)k * d);
        blob.insert(blob.end(), desc->logc, desc->logc + k);
        for (int c = 0; c < k; ++c) {
            double ss = 0.0;
            for (int i = 0; i < d; ++i) ss += desc->mu[(size_t)c * d + i] * desc->mu[(size_t)c * d + i];
            scale = std::fmax(scale, std::sqrt(ss));
        }
        break;
    case GSSS_BINGHAM:
        if (!desc->A) {
            set_error("Bingham needs A");
            return GSSS_E_INVALID;
        }
        blob.assign(desc->A, desc->A + (size_t)d * d);
        if (desc->mu)  // BinghamFisher: linear term b
            blob.insert(blob.end(), desc->mu, desc->mu + d);
        else
            blob.insert(blob.end(), (size_t)d, 0.0);
        {
            bool diag = true;
            for (int i = 0; i < d && diag; ++i)
                for (int j = 0; j < d; ++j)
                    if (i != j && desc->A[(size_t)i * d + j] != 0.0) {
                        diag = false;
                        break;
                    }
            bingham_diagonal = diag;
        }
        break;
    case GSSS_CURVE_VMF: {
        if (k < 2 || !desc->knots) {
            set_error("curve-vMF needs k >= 2 knots");
            return GSSS_E_INVALID;
        }
        blob.assign(desc->knots, desc->knots + (size_t)k * d);
        // per segment: theta = distance(a, b) and its cos / sin / sin + 1e-10 -- the x-independent
        // quantities distance_slerp recomputes on every call (spherical_curve.py:28-31)
        for (int s = 0; s + 1 < k; ++s) {
            const double *a = desc->knots + (size_t)s * d, *b = a + d;
            double ab = 0.0;
            for (int i = 0; i < d; ++i) ab += a[i] * b[i];
            ab = ab < -1.0 ? -1.0 : (ab > 1.0 ? 1.0 : ab);
            const double theta = std::acos(ab);
            blob.push_back(theta);
            blob.push_back(std::cos(theta));
            blob.push_back(std::sin(theta));
            blob.push_back(std::sin(theta) + 1e-10);
        }
        break;
    }
    case GSSS_CPD: {
        const int ns = k, nt = desc->n_target, dt = desc->target_dim, kn = desc->k_nn;
        if (d != 4 || ns < 1 || ns > 65535 || nt < 1 || nt > 4000 || (dt != 2 && dt != 3) || kn < 1 || kn > 24 || kn > ns ||
            !desc->source || !desc->source_w || !desc->target || !desc->target_w || !(desc->sigma > 0.0)) {
            set_error("registration target needs d = 4, 1 <= k_nn <= min(24, source points), target_dim 2 or 3, sigma > 0, "
                      "<= 65535 source and <= 4000 target points, and all four arrays");
            return GSSS_E_INVALID;
        }
        if ((size_t)(4 * ns + 4 * nt + 8) * sizeof(double) > 150 * 1024) {
            set_error("registration target: the point clouds do not fit the LDS");
            return GSSS_E_UNSUPPORTED;
        }
        const double s2 = desc->sigma * desc->sigma;
        // registration.py:215-219 (CoherentPointDrift) / :110-111 (GaussianMixtureModel)
        const double log_const = (desc->outlier ? std::log(1.0 - desc->omega) : 0.0) - 0.5 * dt * std::log(2.0 * 3.141592653589793 * s2);
        blob.assign(desc->source, desc->source + (size_t)3 * ns);
        bool uniform = true;
        for (int i = 0; i < ns; ++i) {
            blob.push_back(std::log(desc->source_w[i]) + log_const);
            uniform = uniform && desc->source_w[i] == desc->source_w[0];
        }
        for (int l = 0; l < nt; ++l)
            for (int j = 0; j < 3; ++j) blob.push_back(j < dt ? desc->target[(size_t)l * dt + j] : 0.0);
        blob.insert(blob.end(), desc->target_w, desc->target_w + nt);
        blob.push_back(std::log(desc->omega + 1e-308) - desc->log_volume);  // registration.py:236
        blob.push_back(0.5 / s2);
        blob.push_back(desc->beta);
        blob.push_back(desc->outlier ? 1.0 : 0.0);
        blob.push_back((double)dt);
        blob.push_back((double)kn);
        blob.push_back(0.0);
        blob.push_back(0.0);
        cpd_variant = (kn <= 8 ? 0 : 2) + (uniform ? 0 : 1);
        break;
    }
    default:
        set_error("unknown target kind %d", desc->kind);
        return GSSS_E_INVALID;
    }
    int ndev = gsss_device_count();
    if (ndev <= 0 || device < 0 || device >= ndev) {
        set_error("device %d not available (%d HIP devices visible)", device, ndev);
        return GSSS_E_NO_DEVICE;
    }
    DeviceGuard guard(device);
    if (!guard.ok) return GSSS_E_HIP;
    gsss_target *t = new (std::nothrow) gsss_target();
    if (!t) {
        set_error("out of host memory");
        return GSSS_E_INVALID;
    }
    t->device = device;
    t->blob_doubles = blob.size();
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&t->blob_dev), blob.size() * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(t->blob_dev, blob.data(), blob.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        set_error("copying target parameters failed: %s", hipGetErrorString(e));
        if (t->blob_dev) (void)hipFree(t->blob_dev);
        delete t;
        return GSSS_E_HIP;
    }
    t->tb.blob = t->blob_dev;
    t->tb.kind = desc->kind;
    t->tb.d = d;
    // Bingham: k holds flags -- bit 0 a diagonal A, bit 1 a linear term (BinghamFisher)
    t->tb.k = desc->kind == GSSS_BINGHAM ? ((bingham_diagonal ? 1 : 0) | (desc->mu ? 2 : 0)) : k;
    if (desc->kind == GSSS_CPD) t->tb.k = k | (desc->n_target << 16);       // registration: both cloud sizes
    t->cpd_variant = cpd_variant;
    t->tb.dpad = 0;
    t->tb.kappa = desc->kappa;
    t->tb.scale = scale;
    *out = t;
    return GSSS_OK;
}

int gsss_target_destroy(gsss_target *t)
{
    if (!t) return GSSS_OK;
    DeviceGuard guard(t->device);
    if (t->blob_dev) (void)hipFree(t->blob_dev);
    delete t;
    return GSSS_OK;
}

int gsss_target_dim(const gsss_target *t) { return t ? t->tb.d : GSSS_E_INVALID; }

static int logprob_or_gradient(const gsss_target *t, const double *x_dev, int64_t n, double *out_dev, bool grad, void *stream)
{
    if (!t || n < 0 || (n > 0 && (!x_dev || !out_dev))) {
        set_error(grad ? "bad argument to gsss_gradient" : "bad argument to gsss_logprob");
        return GSSS_E_INVALID;
    }
    if (n == 0) return GSSS_OK;
    const int vec = select_vec_for(t->tb, 0);
    if (vec < 0) return vec;
    DeviceGuard guard(t->device);
    if (!guard.ok) return GSSS_E_HIP;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (t->tb.kind) {
    case GSSS_VMF_MIXTURE: return launch_logprob<VmfMixture>(vec, t->tb, x_dev, n, out_dev, grad, st);
    case GSSS_BINGHAM: return launch_logprob<Bingham>(vec, t->tb, x_dev, n, out_dev, grad, st);
    case GSSS_CURVE_VMF: return launch_logprob<CurveVmf>(vec, t->tb, x_dev, n, out_dev, grad, st);
    case GSSS_CPD: return launch_cpd_logprob(t->cpd_variant, t->tb, x_dev, n, out_dev, grad, st);
    }
    set_error("corrupt target");
    return GSSS_E_INVALID;
}

int gsss_logprob(const gsss_target *t, const double *x_dev, int64_t n, double *out_dev, void *stream)
{
    return logprob_or_gradient(t, x_dev, n, out_dev, false, stream);
}

int gsss_gradient(const gsss_target *t, const double *x_dev, int64_t n, double *grad_dev, void *stream)
{
    return logprob_or_gradient(t, x_dev, n, grad_dev, true, stream);
}

int gsss_run(const gsss_target *t, const gsss_run_args *a, void *stream)
{
    if (!t || !a) {
        set_error("null argument");
        return GSSS_E_INVALID;
    }
    if (a->n_chains < 0 || a->n_steps < 0 || a->thin < 1 || a->max_tries < 1) {
        set_error("need n_chains >= 0, n_steps >= 0, thin >= 1, max_tries >= 1");
        return GSSS_E_INVALID;
    }
    const bool mh = a->sampler == GSSS_RWMH || a->sampler == GSSS_HMC || a->sampler == GSSS_INDEP || a->sampler == GSSS_MIX;
    if (a->sampler != GSSS_SHRINK && a->sampler != GSSS_REJECT && !mh) {
        set_error("unknown sampler %d", a->sampler);
        return GSSS_E_INVALID;
    }
    if (mh && (a->mode != GSSS_MODE_EXACT || !a->stepsize_dev || a->adapt_steps < 0 || a->stats_dev ||
               (a->sampler == GSSS_HMC && a->n_leapfrog < 1))) {
        set_error("GSSS_RWMH / GSSS_HMC / GSSS_INDEP / GSSS_MIX need GSSS_MODE_EXACT, stepsize_dev, adapt_steps >= 0, no stats_dev "
                  "(HMC: n_leapfrog >= 1)");
        return GSSS_E_INVALID;
    }
    if (a->sampler == GSSS_MIX && (!(a->mixing_probability >= 0.0 && a->mixing_probability <= 1.0) || !a->adapt_left_dev)) {
        set_error("GSSS_MIX needs 0 <= mixing_probability <= 1 and adapt_left_dev");
        return GSSS_E_INVALID;
    }
    if (a->mode != GSSS_MODE_EXACT && a->mode != GSSS_MODE_FAST) {
        set_error("unknown mode %d", a->mode);
        return GSSS_E_INVALID;
    }
    if (a->chain_offset + (uint64_t)a->n_chains > (1ull << 48) || a->step_offset + (uint64_t)a->n_steps >= kInitStep) {
        set_error("chain / step ids exceed the 48-bit counter space");
        return GSSS_E_INVALID;
    }
    last_launch() = LaunchInfo{0, 0, 0.0};
    if (a->n_chains == 0) return GSSS_OK;
    if (!a->state_dev) {
        set_error("state_dev is null");
        return GSSS_E_INVALID;
    }
    if (a->samples_chain_rows < 0 || (a->samples_chain_rows > 0 && a->samples_chain_rows < a->n_steps / a->thin)) {
        set_error("samples_chain_rows must be 0 or >= the rows this call writes");
        return GSSS_E_INVALID;
    }
    if (a->replay_dev && a->replay_stride < 1) {
        set_error("replay_stride must be >= 1");
        return GSSS_E_INVALID;
    }
    if (a->mode == GSSS_MODE_FAST && a->variant != 0 && a->variant != GSSS_VARIANT_FAST_DOUBLE && a->variant != GSSS_VARIANT_FAST_VERIFY) {
        set_error("fast mode takes variant 0, GSSS_VARIANT_FAST_DOUBLE or GSSS_VARIANT_FAST_VERIFY");
        return GSSS_E_INVALID;
    }
    const int vec = a->mode == GSSS_MODE_FAST ? 0 : select_vec_for(t->tb, a->variant);
    if (vec < 0) return vec;
    RunBlock rb{};
    rb.state = a->state_dev;
    rb.samples = a->samples_dev;
    rb.n_reject = a->n_reject_dev;
    rb.n_tries = a->n_tries_dev;
    rb.err = a->err_dev;
    rb.replay = a->replay_dev;
    rb.rng_state = a->rng_state_dev;
    rb.replay_stride = a->replay_stride;
    rb.n_chains = a->n_chains;
    rb.n_steps = a->n_steps;
    rb.thin = a->thin;
    rb.seed = a->seed;
    rb.chain_offset = a->chain_offset;
    rb.step_offset = a->step_offset;
    rb.sampler = a->sampler;
    rb.max_tries = a->max_tries;
    rb.keep_rows = a->samples_chain_rows;
    if (a->placement < 0 || a->placement > 2) {
        set_error("placement must be 0 (auto), 1 (packed) or 2 (spread)");
        return GSSS_E_INVALID;
    }
    // The library's choice: one wavefront per chain while that beats the packed throughput kernels (tools/bench_placement.py,
    // fast mode, chain-steps/s packed (one chain per lane at these sizes) | spread: README mixture 3072 chains 8.3e8 | 9.0e8,
    // 4096 1.11e9 | 0.95e9; K = 10 mixture 2048 3.7e8 | 4.4e8, 3072 5.5e8 | 4.5e8; Bingham d = 10 3072 6.9e8 | 8.0e8, 4096
    // 9.2e8 | 8.4e8; curve d = 10 (group kernel) 1024 chains 1.3e8 | 2.0e8, 2048 2.6e8 | 2.0e8).
    const int64_t spread_max = a->mode != GSSS_MODE_FAST ? 2048 : (t->tb.kind == GSSS_CURVE_VMF ? 1536 : 3072);
    rb.spread = a->placement == 2 || (a->placement == 0 && a->n_chains <= spread_max) ? 1 : 0;
    rb.screen = (a->mode == GSSS_MODE_FAST && a->variant == GSSS_VARIANT_FAST_DOUBLE) ? 0 : (a->variant == GSSS_VARIANT_FAST_VERIFY ? 2 : 1);
    rb.stats = a->stats_dev;
    rb.stats_dirs = a->stats_dirs_dev;
    rb.stats_lags = a->stats_lags;
    rb.stats_modes = a->stats_modes;
    rb.stats_flags = a->stats_flags;
    if (a->stats_dev != nullptr) {
        if (!a->stats_dirs_dev || a->stats_lags < 0 || a->stats_modes < 0 || a->stats_lags > 4096 || a->stats_modes > 4096 ||
            (a->stats_flags & ~GSSS_STATS_NO_SECOND_MOMENT)) {
            set_error("stats_dev needs stats_dirs_dev, 0 <= stats_lags <= 4096, 0 <= stats_modes <= 4096 and known stats_flags");
            return GSSS_E_INVALID;
        }
        // the lane-group layouts form the d (d + 1) / 2 second-moment sums through cross-lane reads: small d only
        if (!(a->stats_flags & GSSS_STATS_NO_SECOND_MOMENT) && t->tb.d > 64) {
            set_error("second moments are accumulated for d <= 64 (d (d + 1) / 2 rows per chain): set GSSS_STATS_NO_SECOND_MOMENT");
            return GSSS_E_UNSUPPORTED;
        }
    }
    DeviceGuard guard(t->device);
    if (!guard.ok) return GSSS_E_HIP;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool replay = a->replay_dev != nullptr;
    if (replay && a->rng_state_dev) {
        set_error("replay_dev and rng_state_dev are mutually exclusive");
        return GSSS_E_INVALID;
    }

    const int draws = replay ? kDrawsReplay : (a->rng_state_dev ? kDrawsNumpy : kDrawsPhilox);
    if (a->mode == GSSS_MODE_FAST && a->rng_state_dev) {  // a generator per chain: the lane-per-chain shapes only
        FastProbe pr;
        if (fast_dispatch(t->tb, rb, false, &pr, nullptr) != GSSS_OK || !pr.lane) {
            set_error("in fast mode the numpy stream is served by the lane-per-chain kernels only (gsss_variant_name: "
                      "\"fast-lane\"); use GSSS_MODE_EXACT");
            return GSSS_E_UNSUPPORTED;
        }
    }
    if (a->mode == GSSS_MODE_FAST) {
        if (a->n_steps > 0x7FFFFFFFll || a->thin > 0x7FFFFFFFll || a->n_chains > 0x7FFFFFFFll - 1024 ||
            a->replay_stride > 0x7FFFFFFFll) {
            set_error("fast mode takes at most 2^31-1 steps / chains per call");
            return GSSS_E_INVALID;
        }
        return fast_dispatch(t->tb, rb, replay, nullptr, st);
    }
    if (mh) {
        if (a->rng_state_dev && (a->sampler == GSSS_RWMH || a->sampler == GSSS_MIX) && t->tb.d < 3) {
            set_error("numpy's gamma(1) is an exponential ziggurat, not restated: RWMH on numpy's stream needs d >= 3");
            return GSSS_E_UNSUPPORTED;
        }
        MhBlock mb;
        mb.stepsize = a->stepsize_dev;
        mb.n_accept = a->n_accept_dev;
        mb.momenta = a->sampler == GSSS_HMC ? a->momenta_dev : nullptr;
        mb.adapt_steps = a->adapt_steps;
        mb.n_leapfrog = a->n_leapfrog;
        mb.kind = a->sampler;
        mb.mix_alpha = a->mixing_probability;
        mb.adapt_left = a->adapt_left_dev;
        mb.n_rwmh = a->n_rwmh_dev;
        mb.momenta_samples = a->sampler == GSSS_HMC ? a->momenta_samples_dev : nullptr;
        mb.stepsize_trace = a->sampler == GSSS_HMC ? nullptr : a->stepsize_trace_dev;
        if (t->tb.kind == GSSS_CPD) return launch_cpd_mh(t->cpd_variant, draws, a->sampler, t->tb, rb, mb, st);
        switch (t->tb.kind) {
        case GSSS_VMF_MIXTURE: return launch_mh<VmfMixture>(vec, draws, a->sampler, t->tb, rb, mb, st);
        case GSSS_BINGHAM: return launch_mh<Bingham>(vec, draws, a->sampler, t->tb, rb, mb, st);
        case GSSS_CURVE_VMF: return launch_mh<CurveVmf>(vec, draws, a->sampler, t->tb, rb, mb, st);
        }
        set_error("corrupt target");
        return GSSS_E_INVALID;
    }
    if (t->tb.kind == GSSS_CPD) {
        if (a->stats_dev) {
            set_error("running statistics are not built for registration targets");
            return GSSS_E_UNSUPPORTED;
        }
        return launch_cpd_run(t->cpd_variant, draws, t->tb, rb, st);
    }
    switch (t->tb.kind) {
    case GSSS_VMF_MIXTURE: return launch_run<VmfMixture>(vec, draws, t->tb, rb, st);
    case GSSS_BINGHAM: return launch_run<Bingham>(vec, draws, t->tb, rb, st);
    case GSSS_CURVE_VMF: return launch_run<CurveVmf>(vec, draws, t->tb, rb, st);
    }
    set_error("corrupt target");
    return GSSS_E_INVALID;
}

int gsss_last_launch(int64_t *grid_out, int32_t *slice_steps_out, double *sliced_fraction_out)
{
    if (grid_out) *grid_out = last_launch().grid;
    if (slice_steps_out) *slice_steps_out = last_launch().slice_steps;
    if (sliced_fraction_out) *sliced_fraction_out = last_launch().sliced_fraction;
    return GSSS_OK;
}

int64_t gsss_stats_rows(int32_t d, int32_t n_modes, int32_t n_lags, int32_t flags)
{
    if (d < 2 || n_modes < 0 || n_lags < 0 || (flags & ~GSSS_STATS_NO_SECOND_MOMENT)) return GSSS_E_INVALID;
    const int64_t second = (flags & GSSS_STATS_NO_SECOND_MOMENT) ? 0 : (int64_t)d * (d + 1) / 2;
    return 1 + 2 * (int64_t)d + second + 2 + n_modes + 2 + 3 * (int64_t)n_lags;
}

int gsss_mode_supported(const gsss_target *t, int32_t mode)
{
    if (!t) return 0;
    if (mode == GSSS_MODE_EXACT) return select_vec(t->tb.d, 0) >= 0;
    if (mode == GSSS_MODE_FAST) {
        FastProbe pr;
        return fast_dispatch(t->tb, RunBlock{}, false, &pr, nullptr) == GSSS_OK;
    }
    return 0;
}

const char *gsss_variant_name(const gsss_target *t, int32_t mode, int32_t variant)
{
    if (!t) return "";
    if (mode == GSSS_MODE_FAST) {
        FastProbe pr;
        if (fast_dispatch(t->tb, RunBlock{}, false, &pr, nullptr) != GSSS_OK) return "";
        return pr.lane ? "fast-lane" : "fast-coop";
    }
    const int vec = select_vec_for(t->tb, variant);
    if (vec < 0) return "";
    int n;
    const VecInfo *v = vec_table(&n);
    for (int i = 0; i < n; ++i)
        if (v[i].id == vec) return v[i].name;
    return "";
}

const char *gsss_kernel_name(const gsss_target *t, int32_t mode, int32_t variant, int32_t placement)
{
    static thread_local char name[200];
    name[0] = 0;
    if (!t) return name;
    const bool spread = placement == 2;
    if (mode == GSSS_MODE_FAST) {
        FastProbe pr;
        RunBlock rbp{};
        rbp.screen = variant == GSSS_VARIANT_FAST_DOUBLE || spread ? 0 : 1;
        if (fast_dispatch(t->tb, rbp, false, &pr, nullptr) != GSSS_OK) return name;
        if (pr.lane && spread && t->tb.d <= 16)
            snprintf(name, sizeof(name), "wave_kernel%s", strchr(pr.name, '<') ? strchr(pr.name, '<') : "");
        else
            snprintf(name, sizeof(name), "%s", pr.name);
        return name;
    }
    const char *vec = gsss_variant_name(t, mode, variant);
    const char *tgt = t->tb.kind == GSSS_VMF_MIXTURE ? "VmfMixture" : (t->tb.kind == GSSS_BINGHAM ? "Bingham" : "CurveVmf");
    if (vec[0]) snprintf(name, sizeof(name), "run_kernel<%s, %s>", vec, tgt);
    return name;
}

int gsss_sample_sphere(uint64_t seed, uint64_t chain_offset, int64_t n, int32_t d, double *state_dev, int device,
                       void *stream)
{
    if (n < 0 || d < 2 || (n > 0 && !state_dev)) {
        set_error("bad argument to gsss_sample_sphere");
        return GSSS_E_INVALID;
    }
    if (n == 0) return GSSS_OK;
    DeviceGuard guard(device);
    if (!guard.ok) return GSSS_E_HIP;
    const int64_t grid = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(sample_sphere_kernel, dim3((unsigned)grid), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       seed, chain_offset, n, (int)d, state_dev);
    GSSS_HIP_TRY(hipGetLastError());
    return GSSS_OK;
}

int gsss_tangent_s2(const double *x_dev, const uint32_t *w_dev, int64_t n, int32_t table_driven, double *out_dev, int device, void *stream)
{
    if (n < 0 || (n > 0 && (!x_dev || !w_dev || !out_dev))) {
        set_error("bad argument to gsss_tangent_s2");
        return GSSS_E_INVALID;
    }
    if (n == 0) return GSSS_OK;
    DeviceGuard guard(device);
    if (!guard.ok) return GSSS_E_HIP;
    const int64_t grid = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(tangent_s2_kernel, dim3((unsigned)grid), dim3(kBlock), 0, static_cast<hipStream_t>(stream), x_dev, w_dev, n,
                       (int)table_driven, out_dev);
    GSSS_HIP_TRY(hipGetLastError());
    return GSSS_OK;
}

int gsss_rows_to_components(const double *in_dev, double *out_dev, int64_t n, int32_t d, int device, void *stream)
{
    return transpose(in_dev, out_dev, n, d, device, static_cast<hipStream_t>(stream));
}

int gsss_components_to_rows(const double *in_dev, double *out_dev, int64_t n, int32_t d, int device, void *stream)
{
    return transpose(in_dev, out_dev, d, n, device, static_cast<hipStream_t>(stream));
}

int gsss_samples_to_chains(const double *in_dev, double *out_dev, int64_t n, int64_t n_keep, int32_t d, int device,
                           void *stream)
{
    // [n_keep][d][n] -> [n][n_keep][d] is the transpose of an [n_keep*d][n] matrix
    return transpose(in_dev, out_dev, n_keep * d, n, device, static_cast<hipStream_t>(stream));
}

int gsss_malloc(void **out_dev, size_t bytes, int device)
{
    if (!out_dev) {
        set_error("null argument");
        return GSSS_E_INVALID;
    }
    DeviceGuard guard(device);
    if (!guard.ok) return GSSS_E_HIP;
    GSSS_HIP_TRY(hipMalloc(out_dev, bytes ? bytes : 1));
    return GSSS_OK;
}

int gsss_free(void *p_dev, int device)
{
    if (!p_dev) return GSSS_OK;
    DeviceGuard guard(device);
    if (!guard.ok) return GSSS_E_HIP;
    GSSS_HIP_TRY(hipFree(p_dev));
    return GSSS_OK;
}

int gsss_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes, int device, void *stream)
{
    DeviceGuard guard(device);
    if (!guard.ok) return GSSS_E_HIP;
    hipStream_t st = static_cast<hipStream_t>(stream);
    GSSS_HIP_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, st));
    GSSS_HIP_TRY(hipStreamSynchronize(st));
    return GSSS_OK;
}

int gsss_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes, int device, void *stream)
{
    DeviceGuard guard(device);
    if (!guard.ok) return GSSS_E_HIP;
    hipStream_t st = static_cast<hipStream_t>(stream);
    GSSS_HIP_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, st));
    GSSS_HIP_TRY(hipStreamSynchronize(st));
    return GSSS_OK;
}

int gsss_malloc_host(void **out_host, size_t bytes, int device)
{
    if (!out_host) {
        set_error("null argument");
        return GSSS_E_INVALID;
    }
    DeviceGuard guard(device);
    if (!guard.ok) return GSSS_E_HIP;
    GSSS_HIP_TRY(hipHostMalloc(out_host, bytes ? bytes : 1, hipHostMallocDefault));
    return GSSS_OK;
}

int gsss_free_host(void *p_host)
{
    if (!p_host) return GSSS_OK;
    GSSS_HIP_TRY(hipHostFree(p_host));
    return GSSS_OK;
}

int gsss_host_register(void *p_host, size_t bytes, int device)
{
    if (!p_host || bytes == 0) {
        set_error("bad argument to gsss_host_register");
        return GSSS_E_INVALID;
    }
    DeviceGuard guard(device);
    if (!guard.ok) return GSSS_E_HIP;
    GSSS_HIP_TRY(hipHostRegister(p_host, bytes, hipHostRegisterDefault));
    return GSSS_OK;
}

int gsss_host_unregister(void *p_host)
{
    if (!p_host) return GSSS_OK;
    GSSS_HIP_TRY(hipHostUnregister(p_host));
    return GSSS_OK;
}

int gsss_memcpy_d2h_async(void *dst_host, const void *src_dev, size_t bytes, int device, void *stream)
{
    DeviceGuard guard(device);
    if (!guard.ok) return GSSS_E_HIP;
    GSSS_HIP_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    return GSSS_OK;
}

int gsss_memset(void *dst_dev, int value, size_t bytes, int device, void *stream)
{
    DeviceGuard guard(device);
    if (!guard.ok) return GSSS_E_HIP;
    GSSS_HIP_TRY(hipMemsetAsync(dst_dev, value, bytes, static_cast<hipStream_t>(stream)));
    return GSSS_OK;
}

int gsss_stream_synchronize(int device, void *stream)
{
    DeviceGuard guard(device);
    if (!guard.ok) return GSSS_E_HIP;
    GSSS_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return GSSS_OK;
}

}  // extern "C"
