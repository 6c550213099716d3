// Input pipeline on device: from decoded uint8 frames to the normalised fp32 NCHW
// tensor the model takes, i.e. the reference's torchvision transforms
//   eval : Grayscale(3) -> Resize((224,224)) -> ToTensor -> Normalize           (train_vit.py:751-756, infer.py:346-353)
//   train: Grayscale(3) -> Resize((246,246)) -> RandomCrop(224) -> ToTensor -> Normalize -> AddGaussianNoise(0.05)
//                                                                                (train_vit.py:722-749, 557-567)
// restated bit-exactly for the integer part: Pillow's rgb2l, then Pillow's two-pass antialiased triangle
// resampling with 22-bit fixed-point coefficients and a uint8 intermediate (libImaging Convert.c / Resample.c).
// The coefficient tables come from the host (vit_amd/pipeline.py); only the crop window is computed.
//
// Both kernels are HBM-bound byte work: pass 1 reads every source byte once (3.56 MB per 1475x804 RGB frame)
// and writes H x 224 bytes, pass 2 reads that and writes 3 x 224 x 224 fp32 (602 KB).
//   pass 1 (preproc_h_kernel): a workgroup takes a run of source rows; the raw row is copied to LDS with aligned
//     dword loads, converted to luma in LDS, then thread x accumulates output column crop_left + x.
//   pass 2 (preproc_v_kernel): thread (y, x) accumulates down a column of the intermediate (coalesced along x),
//     converts to float, normalises per channel, adds counter-based Gaussian noise, writes three planes.
#include <mutex>

#include "common.h"
#include "rvit_internal.h"

namespace {

constexpr int PP_PREC = 22;   // 32 - 8 - 2, Resample.c PRECISION_BITS
constexpr int PP_ROWS = 8;    // source rows per workgroup in pass 1

__device__ __forceinline__ int pp_clip8(int v) {
  v >>= PP_PREC;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

struct __attribute__((packed, aligned(4))) PPQuad { unsigned x, y, z, w; };   // 16 bytes at 4-byte alignment

// 16 bytes starting at the 4-byte aligned offset `off`; the tail of the buffer is assembled from what exists
__device__ __forceinline__ PPQuad pp_load16(const uint8_t* __restrict__ base, long off, long total) {
  PPQuad q;
  if (off + 16 <= total) {
    q = *reinterpret_cast<const PPQuad*>(base + off);
  } else {
    unsigned d[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      d[i] = 0;
      const long o = off + 4 * i;
      if (o + 4 <= total) d[i] = *reinterpret_cast<const unsigned*>(base + o);
      else
        for (int e = 0; e < 4; ++e)
          if (o + e < total) d[i] |= (unsigned)base[o + e] << (8 * e);
    }
    q.x = d[0]; q.y = d[1]; q.z = d[2]; q.w = d[3];
  }
  return q;
}

__device__ __forceinline__ unsigned pp_l(unsigned r, unsigned g, unsigned b) {   // Convert.c rgb2l
  return (r * 19595u + g * 38470u + b * 7471u + 0x8000u) >> 16;
}

// Pass 1.  A workgroup owns ROWS source rows of one frame.  Phase 1: every thread converts groups of four
// pixels to luma straight from (dword-aligned, 16-byte) global loads -- rows of W*C bytes start at any byte
// offset, v_alignbyte realigns -- and stores one dword of luma to LDS.  Phase 2: thread x keeps the taps of
// output column crop_left + x in flight once for all ROWS rows.
template <int ROWS>
__global__ __launch_bounds__(256) void preproc_h_kernel(const uint8_t* __restrict__ img, long total_bytes, int H, int W,
                                                        int C, const int* __restrict__ hcoef,
                                                        const int* __restrict__ hbounds, int hk,
                                                        const int* __restrict__ crop, int out_side,
                                                        uint8_t* __restrict__ tmp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char luma[];   // [ROWS][Wp]
  const int b = blockIdx.y, r0 = blockIdx.x * ROWS;
  const int nrows = H - r0 < ROWS ? H - r0 : ROWS;
  const int G = (W + 3) >> 2, Wp = G * 4;
  const long row_bytes = (long)W * C;
  for (int idx = threadIdx.x; idx < nrows * G; idx += 256) {
    const int rr = idx / G, g = idx - rr * G;
    const long off = ((long)b * H + r0 + rr) * row_bytes + (long)g * 4 * C;
    const long al = off & ~3L;
    const int sh = (int)(off & 3);
    unsigned out4;
    if (C == 3) {
      const PPQuad q = pp_load16(img, al, total_bytes);
      const unsigned w0 = __builtin_amdgcn_alignbyte(q.y, q.x, sh), w1 = __builtin_amdgcn_alignbyte(q.z, q.y, sh),
                     w2 = __builtin_amdgcn_alignbyte(q.w, q.z, sh);
      // bytes: w0 = R0 G0 B0 R1, w1 = G1 B1 R2 G2, w2 = B2 R3 G3 B3 (little endian)
      const unsigned l0 = pp_l(w0 & 255, (w0 >> 8) & 255, (w0 >> 16) & 255);
      const unsigned l1 = pp_l(w0 >> 24, w1 & 255, (w1 >> 8) & 255);
      const unsigned l2 = pp_l((w1 >> 16) & 255, w1 >> 24, w2 & 255);
      const unsigned l3 = pp_l((w2 >> 8) & 255, (w2 >> 16) & 255, w2 >> 24);
      out4 = l0 | (l1 << 8) | (l2 << 16) | (l3 << 24);
    } else if (C == 4) {
      const PPQuad q = pp_load16(img, al, total_bytes);   // rows of 4-byte pixels are dword aligned: sh == 0
      out4 = pp_l(q.x & 255, (q.x >> 8) & 255, (q.x >> 16) & 255) | (pp_l(q.y & 255, (q.y >> 8) & 255, (q.y >> 16) & 255) << 8) |
             (pp_l(q.z & 255, (q.z >> 8) & 255, (q.z >> 16) & 255) << 16) | (pp_l(q.w & 255, (q.w >> 8) & 255, (q.w >> 16) & 255) << 24);
    } else {
      unsigned d0 = 0, d1 = 0;
      if (al + 8 <= total_bytes) {
        d0 = *reinterpret_cast<const unsigned*>(img + al);
        d1 = *reinterpret_cast<const unsigned*>(img + al + 4);
      } else {
        for (int e = 0; e < 8; ++e)
          if (al + e < total_bytes) (e < 4 ? d0 : d1) |= (unsigned)img[al + e] << (8 * (e & 3));
      }
      out4 = __builtin_amdgcn_alignbyte(d1, d0, sh);
    }
    *reinterpret_cast<unsigned*>(luma + rr * Wp + g * 4) = out4;
  }
  __syncthreads();
  const int x = threadIdx.x;
  if (x >= out_side) return;
  const int cx = crop ? crop[b * 2 + 1] : 0;
  const int xmin = hbounds[(cx + x) * 2], n = hbounds[(cx + x) * 2 + 1];
  const int* k = hcoef + (long)(cx + x) * hk;
  int acc[ROWS];
#pragma unroll
  for (int rr = 0; rr < ROWS; ++rr) acc[rr] = 1 << (PP_PREC - 1);
  for (int j = 0; j < n; ++j) {
    const int kj = k[j];
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) acc[rr] += (int)luma[rr * Wp + xmin + j] * kj;
  }
#pragma unroll
  for (int rr = 0; rr < ROWS; ++rr)
    if (rr < nrows) tmp[((long)b * H + r0 + rr) * out_side + x] = (uint8_t)pp_clip8(acc[rr]);
}

__device__ __forceinline__ float pp_gauss(uint32_t k0, uint32_t k1, uint32_t idx) {
  // Box-Muller on two 24-bit uniforms from the same double-mix counter hash as the dropout masks
  const uint32_t a = rvit_mix32(rvit_mix32(idx ^ k0) + k1);
  const uint32_t c = rvit_mix32(a ^ 0x9E3779B9u);
  const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0, 1]
  const float u2 = (float)(c >> 8) * (1.0f / 16777216.0f);            // [0, 1)
  return sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530718f * u2);
}

// Pass 2.  A thread owns four adjacent output pixels of one row: per tap one dword of the intermediate (four
// columns), four accumulators, then three float4 stores (one per channel plane).
__global__ __launch_bounds__(256) void preproc_v_kernel(const uint8_t* __restrict__ tmp, int H,
                                                        const int* __restrict__ vcoef, const int* __restrict__ vbounds,
                                                        int vk, const int* __restrict__ crop, int out_side,
                                                        float noise_std, uint32_t k0, uint32_t k1,
                                                        float* __restrict__ out) {
  const int b = blockIdx.y;
  const int q = out_side >> 2;                       // quads per row
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= out_side * q) return;
  const int y = idx / q, x = (idx - y * q) * 4;
  const int cy = crop ? crop[b * 2] : 0;
  const int ymin = vbounds[(cy + y) * 2], n = vbounds[(cy + y) * 2 + 1];
  const int* k = vcoef + (long)(cy + y) * vk;
  const uint8_t* col = tmp + ((long)b * H + ymin) * out_side + x;
  int acc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = 1 << (PP_PREC - 1);
  for (int j = 0; j < n; ++j) {
    const unsigned d = *reinterpret_cast<const unsigned*>(col + (long)j * out_side);
    const int kj = k[j];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += (int)((d >> (8 * e)) & 255u) * kj;
  }
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (float)pp_clip8(acc[e]) / 255.0f;   // ToTensor
  const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
  const long plane = (long)out_side * out_side;
  const int pix = y * out_side + x;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = (v[e] - mean[c]) / stdv[c];                                 // Normalize
      if (noise_std > 0.f) o[e] += noise_std * pp_gauss(k0, k1, (uint32_t)(((long)b * 3 + c) * plane + pix + e));
    }
    *reinterpret_cast<float4*>(out + ((long)b * 3 + c) * plane + pix) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

}  // namespace

int rvit_preprocess_impl(const uint8_t* img, int B, int H, int W, int C, const int* hcoef, const int* hbounds, int hk,
                         const int* vcoef, const int* vbounds, int vk, int resized, const int* crop, int out_side,
                         uint8_t* tmp, float noise_std, uint64_t noise_seed, float* out, hipStream_t st) {
  RVIT_CHECK_ARG(img && hcoef && hbounds && vcoef && vbounds && tmp && out, "preprocess: null pointer");
  RVIT_CHECK_ARG(B > 0 && H > 0 && W > 0 && (C == 1 || C == 3 || C == 4), "preprocess: bad frame shape B=%d H=%d W=%d C=%d", B, H, W, C);
  RVIT_CHECK_ARG(out_side > 0 && out_side <= 256 && (out_side & 3) == 0 && resized >= out_side && hk > 0 && vk > 0,
                 "preprocess: bad sizes (out_side must be a multiple of 4, at most 256)");
  RVIT_CHECK_ARG(rvit_aligned16(out) && ((uintptr_t)tmp & 3) == 0, "preprocess: out must be 16-byte, tmp 4-byte aligned");
  RVIT_CHECK_ARG(crop || resized == out_side, "preprocess: crop offsets are required when resized != out_side");
  RVIT_CHECK_ARG(noise_std >= 0.f, "preprocess: negative noise_std");
  RVIT_CHECK_ARG((long)B * 3 * out_side * out_side < (1L << 32), "preprocess: batch too large for the noise counter");
  const size_t lds = (size_t)PP_ROWS * (size_t)((W + 3) & ~3);
  RVIT_CHECK_ARG(lds <= 64 * 1024, "preprocess: rows of %d pixels do not fit the LDS luma buffer", W);
  // pass 1 reads 16 bytes at dword-aligned offsets; the last bytes of the buffer are loaded piecewise
  RVIT_CHECK_ARG(((uintptr_t)img & 3) == 0, "preprocess: frame buffer must be 4-byte aligned");
  const DropKey key = make_drop_key(0.f, noise_seed);
  hipLaunchKernelGGL(preproc_h_kernel<PP_ROWS>, dim3((H + PP_ROWS - 1) / PP_ROWS, B), dim3(256), lds, st, img,
                     (long)B * H * W * C, H, W, C, hcoef, hbounds, hk, crop, out_side, tmp);
  RVIT_CHECK_LAUNCH("preprocess(h)");
  hipLaunchKernelGGL(preproc_v_kernel, dim3((out_side * (out_side / 4) + 255) / 256, B), dim3(256), 0, st, tmp, H, vcoef,
                     vbounds, vk, crop, out_side, noise_std, key.k0, key.k1, out);
  RVIT_CHECK_LAUNCH("preprocess(v)");
  return RVIT_OK;
}

// ---- the entry point that allocates nothing (rvit_preprocess_u8_static, include/rvit.h) -------------------------
// Same two kernels; the uint8 intermediate [B][H][out_side] lives in a file-scope device array instead of a
// caller-provided buffer.  Every GPU the library runs on holds its own copy of the array (until the process ends)
// at its own address: `scratch` below is only the host-side handle of the symbol, never a device address.  The
// address on a device is resolved once, with that device current, and kept in pp_scratch_ptr[device]; host code
// uses that pointer wherever the array's memory is meant.
static __device__ __align__(256) uint8_t scratch[RVIT_PREPROCESS_STATIC_MAX_BYTES];

namespace {
constexpr int PP_MAX_DEVICES = 64;
std::mutex pp_mu;
uint8_t* pp_scratch_ptr[PP_MAX_DEVICES];   // guarded by pp_mu

int pp_scratch_of_current_device(hipStream_t st, uint8_t** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= PP_MAX_DEVICES) {
    rvit_set_error("preprocess(static): hipGetDevice failed or device index %d out of range: %s", dev, hipGetErrorString(e));
    return RVIT_ELAUNCH;
  }
  if (st != nullptr) {   // the array of the CURRENT device is used: a stream of another device would fault
    hipDevice_t sdev = -1;
    e = hipStreamGetDevice(st, &sdev);
    if (e != hipSuccess || (int)sdev != dev) {
      rvit_set_error("preprocess(static): the stream belongs to device %d, the current device is %d", (int)sdev, dev);
      return RVIT_EINVAL;
    }
  }
  std::lock_guard<std::mutex> lk(pp_mu);
  if (pp_scratch_ptr[dev] == nullptr) {
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || p == nullptr) {
      rvit_set_error("preprocess(static): hipGetSymbolAddress(scratch) failed on device %d: %s", dev, hipGetErrorString(e));
      return RVIT_ELAUNCH;
    }
    pp_scratch_ptr[dev] = static_cast<uint8_t*>(p);
  }
  *out = pp_scratch_ptr[dev];
  return RVIT_OK;
}
}  // namespace

int rvit_preprocess_static_impl(const uint8_t* img, int B, int H, int W, int C, const int* hcoef, const int* hbounds,
                                int hk, const int* vcoef, const int* vbounds, int vk, int resized, const int* crop,
                                int out_side, float noise_std, uint64_t noise_seed, float* out, hipStream_t st) {
  RVIT_CHECK_ARG(B > 0 && H > 0 && out_side > 0, "preprocess(static): bad shape B=%d H=%d out_side=%d", B, H, out_side);
  const long need = (long)B * H * out_side;
  RVIT_CHECK_ARG(need <= (long)RVIT_PREPROCESS_STATIC_MAX_BYTES,
                 "preprocess(static): B*H*out_side = %ld bytes of intermediate exceed the static scratch (%ld bytes); "
                 "split the batch or call rvit_preprocess_u8 with a caller-owned buffer",
                 need, (long)RVIT_PREPROCESS_STATIC_MAX_BYTES);
  uint8_t* tmp = nullptr;
  const int rc = pp_scratch_of_current_device(st, &tmp);
  if (rc != RVIT_OK) return rc;
  // pass 1 writes every byte of tmp[0, B*H*out_side) that pass 2 reads: nothing has to be cleared between calls
  return rvit_preprocess_impl(img, B, H, W, C, hcoef, hbounds, hk, vcoef, vbounds, vk, resized, crop, out_side, tmp,
                              noise_std, noise_seed, out, st);
}
