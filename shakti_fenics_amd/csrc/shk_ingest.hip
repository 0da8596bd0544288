// Setup-time data ingestion on the device: the two per-node loops of the reference's
// model container, which become visible at 10M nodes.
//   shk_interp_regular_grid  bilinear regular-grid -> nodes, /root/reference/source/model_setup.py:74-91
//                            (scipy RegularGridInterpolator, method "linear", bounds_error=False,
//                            fill_value=None: points outside the grid extrapolate from the edge cell)
//   shk_points_in_polygon    lake outline mask, model_setup.py:68-72 (an O(Nv) Python loop over shapely
//                            `contains` there); even-odd crossing rule here
// Both are context-free: host arrays in, host array out, in the caller's node order.  The arithmetic is
// written in the operation order scipy / the NumPy oracle use and this file is compiled with
// -ffp-contract=off (HIP's __dmul_rn / __dadd_rn are plain operators that the compiler would otherwise fuse
// into FMAs), so results match them bit for bit.
#include <algorithm>
#include <string>
#include <vector>

#include "shk_device.h"

namespace shk {

// index i with g[i] <= x < g[i+1], clamped to [0, n-2] (scipy's find_interval_ascending with extrapolate=1:
// x below the grid -> 0, x at or above the last point -> n-2)
__device__ __forceinline__ int find_interval(const double* __restrict__ g, int n, double x) {
    int lo = 0, hi = n;            // first index with g[idx] > x
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (g[mid] > x) hi = mid; else lo = mid + 1;
    }
    return min(max(lo - 1, 0), n - 2);
}

// f is (nx, ny) row-major, i.e. f[ix * ny + iy] -- the transposed array the reference hands to scipy.
__global__ __launch_bounds__(kBlock) void k_interp_grid(int64_t npts, const double* __restrict__ px,
                                                        const double* __restrict__ py, int nx, int ny,
                                                        const double* __restrict__ xg, const double* __restrict__ yg,
                                                        const double* __restrict__ f, int pairwise,
                                                        double* __restrict__ out) {
    for (int64_t p = blockIdx.x * (int64_t)kBlock + threadIdx.x; p < npts; p += (int64_t)gridDim.x * kBlock) {
        const double x = px[p], y = py[p];
        if (!(x == x) || !(y == y)) { out[p] = __builtin_nan(""); continue; }   // scipy: NaN in, NaN out
        const int i = find_interval(xg, nx, x), j = find_interval(yg, ny, y);
        const double tx = __ddiv_rn(__dsub_rn(x, xg[i]), __dsub_rn(xg[i + 1], xg[i]));
        const double ty = __ddiv_rn(__dsub_rn(y, yg[j]), __dsub_rn(yg[j + 1], yg[j]));
        const double sx = __dsub_rn(1.0, tx), sy = __dsub_rn(1.0, ty);
        const double* f0 = f + (size_t)i * ny + j;
        // corner order (i,j), (i,j+1), (i+1,j), (i+1,j+1), summed left to right.  scipy has two evaluators that
        // round differently: its 2-D float64 fast path (evaluate_linear_2d) forms (f * wx) * wy, the generic
        // _evaluate_linear (float32 / read-only data ...) forms f * (wx * wy).
        double v;
        if (pairwise) {
            v = __dmul_rn(f0[0], __dmul_rn(sx, sy));
            v = __dadd_rn(v, __dmul_rn(f0[1], __dmul_rn(sx, ty)));
            v = __dadd_rn(v, __dmul_rn(f0[ny], __dmul_rn(tx, sy)));
            v = __dadd_rn(v, __dmul_rn(f0[ny + 1], __dmul_rn(tx, ty)));
        } else {
            v = __dmul_rn(__dmul_rn(f0[0], sx), sy);
            v = __dadd_rn(v, __dmul_rn(__dmul_rn(f0[1], sx), ty));
            v = __dadd_rn(v, __dmul_rn(__dmul_rn(f0[ny], tx), sy));
            v = __dadd_rn(v, __dmul_rn(__dmul_rn(f0[ny + 1], tx), ty));
        }
        out[p] = v;
    }
}

// Even-odd rule; edge k runs from (a,b) = poly[k] to (c,d) = poly[k+1 mod m].  Edges are staged through LDS in
// tiles, every thread owns one point.  hit = ((b > y) != (d > y)) and x < (c - a) * (y - b) / (d - b) + a.
constexpr int kPolyTile = 1024;
__global__ __launch_bounds__(kBlock) void k_points_in_polygon(int64_t npts, const double* __restrict__ px,
                                                              const double* __restrict__ py, int m,
                                                              const double2* __restrict__ poly,
                                                              double* __restrict__ out) {
    __shared__ double2 e0[kPolyTile], e1[kPolyTile];
    const int64_t nblk_pts = ((npts + kBlock - 1) / kBlock);
    for (int64_t blk = blockIdx.x; blk < nblk_pts; blk += gridDim.x) {   // uniform trip count per workgroup
        const int64_t p = blk * kBlock + threadIdx.x;
        const double x = p < npts ? px[p] : 0.0, y = p < npts ? py[p] : 0.0;
        bool inside = false;
        for (int k0 = 0; k0 < m; k0 += kPolyTile) {
            const int nt = min(kPolyTile, m - k0);
            __syncthreads();
            for (int k = threadIdx.x; k < nt; k += kBlock) {
                e0[k] = poly[k0 + k];
                e1[k] = poly[(k0 + k + 1) % m];
            }
            __syncthreads();
            for (int k = 0; k < nt; ++k) {
                const double a = e0[k].x, b = e0[k].y, c = e1[k].x, d = e1[k].y;
                if (b == d) continue;
                if ((b > y) != (d > y)) {
                    const double xc = __dadd_rn(__ddiv_rn(__dmul_rn(__dsub_rn(c, a), __dsub_rn(y, b)), __dsub_rn(d, b)), a);
                    if (x < xc) inside = !inside;
                }
            }
        }
        if (p < npts) out[p] = inside ? 1.0 : 0.0;
    }
}

struct DevBuf {   // scope-bound device allocation
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 8)); }
    template <class T> T* as() { return reinterpret_cast<T*>(p); }
};

static int hip_fail(hipError_t e, const char* what) {
    return set_error(std::string(what) + " failed: " + hipGetErrorString(e));
}

}  // namespace shk

using namespace shk;

extern "C" {

int shk_interp_regular_grid(int device, int64_t npts, const double* px, const double* py, int64_t nx, int64_t ny,
                            const double* xg, const double* yg, const double* f_xy, int32_t pairwise_weights,
                            double* out) {
    if (npts < 0 || (npts > 0 && (!px || !py || !out))) return set_error("null point arrays");
    if (nx < 2 || ny < 2 || nx > (1 << 30) || ny > (1 << 30)) return set_error("grid needs at least 2 points per axis");
    if (!xg || !yg || !f_xy) return set_error("null grid arrays");
    for (int64_t i = 1; i < nx; ++i) if (!(xg[i] > xg[i - 1])) return set_error("x grid must be strictly ascending");
    for (int64_t j = 1; j < ny; ++j) if (!(yg[j] > yg[j - 1])) return set_error("y grid must be strictly ascending");
    if (npts == 0) return 0;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || device < 0 || device >= ndev) return set_error("device_id out of range: no such GPU");
    if ((e = hipSetDevice(device)) != hipSuccess) return hip_fail(e, "hipSetDevice");
    DevBuf dpx, dpy, dxg, dyg, df, dout;
    const size_t pb = (size_t)npts * sizeof(double), fb = (size_t)nx * ny * sizeof(double);
    if ((e = dpx.alloc(pb)) != hipSuccess || (e = dpy.alloc(pb)) != hipSuccess || (e = dout.alloc(pb)) != hipSuccess ||
        (e = dxg.alloc(nx * sizeof(double))) != hipSuccess || (e = dyg.alloc(ny * sizeof(double))) != hipSuccess ||
        (e = df.alloc(fb)) != hipSuccess)
        return hip_fail(e, "hipMalloc");
    if ((e = hipMemcpy(dpx.p, px, pb, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(dpy.p, py, pb, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(dxg.p, xg, nx * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(dyg.p, yg, ny * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(df.p, f_xy, fb, hipMemcpyHostToDevice)) != hipSuccess)
        return hip_fail(e, "hipMemcpy (host to device)");
    const int grid = (int)std::min<int64_t>((npts + kBlock - 1) / kBlock, 8192);
    hipLaunchKernelGGL(k_interp_grid, dim3(grid), dim3(kBlock), 0, 0, npts, dpx.as<double>(), dpy.as<double>(), (int)nx,
                       (int)ny, dxg.as<double>(), dyg.as<double>(), df.as<double>(), pairwise_weights ? 1 : 0, dout.as<double>());
    if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_interp_grid launch");
    if ((e = hipMemcpy(out, dout.p, pb, hipMemcpyDeviceToHost)) != hipSuccess) return hip_fail(e, "hipMemcpy (device to host)");
    return 0;
}

int shk_points_in_polygon(int device, int64_t npts, const double* px, const double* py, int64_t m,
                          const double* poly_xy, double* out) {
    if (npts < 0 || (npts > 0 && (!px || !py || !out))) return set_error("null point arrays");
    if (m < 3 || m > (1 << 30) || !poly_xy) return set_error("polygon needs at least 3 vertices");
    if (npts == 0) return 0;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || device < 0 || device >= ndev) return set_error("device_id out of range: no such GPU");
    if ((e = hipSetDevice(device)) != hipSuccess) return hip_fail(e, "hipSetDevice");
    DevBuf dpx, dpy, dpoly, dout;
    const size_t pb = (size_t)npts * sizeof(double), mb = (size_t)m * 2 * sizeof(double);
    if ((e = dpx.alloc(pb)) != hipSuccess || (e = dpy.alloc(pb)) != hipSuccess || (e = dout.alloc(pb)) != hipSuccess ||
        (e = dpoly.alloc(mb)) != hipSuccess)
        return hip_fail(e, "hipMalloc");
    if ((e = hipMemcpy(dpx.p, px, pb, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(dpy.p, py, pb, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(dpoly.p, poly_xy, mb, hipMemcpyHostToDevice)) != hipSuccess)
        return hip_fail(e, "hipMemcpy (host to device)");
    const int grid = (int)std::min<int64_t>((npts + kBlock - 1) / kBlock, 8192);
    hipLaunchKernelGGL(k_points_in_polygon, dim3(grid), dim3(kBlock), 0, 0, npts, dpx.as<double>(), dpy.as<double>(),
                       (int)m, dpoly.as<double2>(), dout.as<double>());
    if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_points_in_polygon launch");
    if ((e = hipMemcpy(out, dout.p, pb, hipMemcpyDeviceToHost)) != hipSuccess) return hip_fail(e, "hipMemcpy (device to host)");
    return 0;
}

}  // extern "C"
