// Corpus index: HBM-resident fp16 rows in the MFMA-fragment tiled layout.
// Stands in for the Milvus collection of the reference
// ("chunking_storing (1).py":14-29 schema/index, :383-396 insert/flush/load).
#include "rf_internal.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <new>

static thread_local char g_err[512] = "";

void rf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* rf_last_error(void) { return g_err; }
extern "C" int rf_version(void) { return 200; }

extern "C" int rf_device_check(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    rf_set_error("no HIP device visible");
    return RF_ERR_DEVICE;
  }
  if (device < 0 || device >= n) {
    rf_set_error("device %d out of range (have %d)", device, n);
    return RF_ERR_DEVICE;
  }
  hipDeviceProp_t prop;
  RF_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    rf_set_error("device %d is %s; this library is built for gfx950 only", device,
                 prop.gcnArchName);
    return RF_ERR_DEVICE;
  }
  return RF_OK;
}

static inline int64_t blocks_for(int64_t rows) { return (rows + RF_BLOCK_ROWS - 1) / RF_BLOCK_ROWS; }

extern "C" size_t rf_index_storage_bytes(int dim, int64_t capacity_rows) {
  if (dim <= 0 || dim % 16 != 0 || capacity_rows <= 0) return 0;
  const size_t KS = (size_t)dim / 16;
  return (size_t)blocks_for(capacity_rows) * KS * RF_FRAG_BYTES + 256;
}

extern "C" int rf_index_create(rf_index_t** out, int dim, int64_t capacity_rows,
                               void* storage_dev, size_t storage_bytes, int device) {
  if (!out || !storage_dev) {
    rf_set_error("rf_index_create: null argument");
    return RF_ERR_INVALID;
  }
  if (!rf_scan_supported_dim(dim)) {
    rf_set_error("rf_index_create: dim %d not supported (need one of 64..1024, multiple of 16, "
                 "with a compiled scan kernel)", dim);
    return RF_ERR_UNSUPPORTED;
  }
  if (capacity_rows <= 0 || capacity_rows > (int64_t)0xFFFFFFE0ll) {
    rf_set_error("rf_index_create: capacity %lld out of range", (long long)capacity_rows);
    return RF_ERR_INVALID;
  }
  const size_t need = rf_index_storage_bytes(dim, capacity_rows);
  if (storage_bytes < need) {
    rf_set_error("rf_index_create: storage %zu B < required %zu B", storage_bytes, need);
    return RF_ERR_CAPACITY;
  }
  if (((uintptr_t)storage_dev & 15) != 0) {
    rf_set_error("rf_index_create: storage not 16-byte aligned");
    return RF_ERR_INVALID;
  }
  int st = rf_device_check(device);
  if (st != RF_OK) return st;
  rf_index* ix = new (std::nothrow) rf_index();
  if (!ix) {
    rf_set_error("out of host memory");
    return RF_ERR_INVALID;
  }
  ix->dim = dim;
  ix->KS = dim / 16;
  ix->device = device;
  ix->capacity = capacity_rows;
  ix->size = 0;
  ix->tiles = (uint4*)storage_dev;
  ix->storage_bytes = storage_bytes;
  ix->max_norm2 = (uint32_t*)((char*)storage_dev + (need - 256));
  {
    hipDeviceProp_t prop;
    RF_HIP(hipGetDeviceProperties(&prop, device));
    ix->num_cus = prop.multiProcessorCount;
  }
  *out = ix;
  return RF_OK;
}

extern "C" int rf_index_destroy(rf_index_t* ix) {
  delete ix;
  return RF_OK;
}

extern "C" int64_t rf_index_size(const rf_index_t* ix) { return ix ? ix->size : -1; }
extern "C" int rf_index_dim(const rf_index_t* ix) { return ix ? ix->dim : -1; }

// ---- kernels ---------------------------------------------------------------

// Zero the tail block(s) that will receive rows and the norm tracker.
__global__ void k_zero_u4(uint4* p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = make_uint4(0, 0, 0, 0);
}

// Row-major fp16 [n, dim] -> tiled.  One thread per 16-byte chunk; consecutive
// threads take consecutive ROWS of the same chunk so the tiled writes are
// contiguous 512-byte runs.
__global__ void k_tile_rows(const uint4* __restrict__ in, uint4* __restrict__ tiles,
                            int64_t first_row, int64_t n, int KS) {
  const int chunks = KS * 2;
  const int64_t total = ((n + 31) / 32) * chunks * 32;  // (rowgroup, chunk, r) index space
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    // i = (rowgroup * chunks + chunk) * 32 + r
    const int r = (int)(i & 31);
    const int64_t t = i >> 5;
    const int chunk = (int)(t % chunks);
    const int64_t rg = t / chunks;
    const int64_t local = rg * 32 + r;
    if (local >= n) continue;
    const uint4 v = in[local * chunks + chunk];
    tiles[rf_chunk_index(first_row + local, chunk, KS)] = v;
  }
}

// max over rows of sum(x^2), tracked as float bits (non-negative floats order
// like unsigned ints).  One wave per row.
__global__ void k_max_norm2(const _Float16* __restrict__ in, int64_t n, int dim,
                            uint32_t* __restrict__ max_norm2) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  float best = 0.f;
  for (int64_t row = wave; row < n; row += nwaves) {
    float s = 0.f;
    for (int d = lane; d < dim; d += 64) {
      const float x = (float)in[row * dim + d];
      s += x * x;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    best = fmaxf(best, s);
  }
  if (lane == 0 && best > 0.f) atomicMax(max_norm2, __builtin_bit_cast(uint32_t, best));
}

__global__ void k_get_rows(const uint4* __restrict__ tiles, const int64_t* __restrict__ rows,
                           int64_t n, int KS, int64_t size, uint4* __restrict__ out) {
  const int chunks = KS * 2;
  const int64_t total = n * chunks;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const int64_t j = i / chunks;
    const int chunk = (int)(i % chunks);
    const int64_t row = rows[j];
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row >= 0 && row < size) v = tiles[rf_chunk_index(row, chunk, KS)];
    out[i] = v;
  }
}

// fp32 row -> (optionally L2-normalised) fp16 row.  One wave per row; the sum
// of squares is accumulated per lane over d = lane, lane+64, ... and combined
// with an xor butterfly (fixed order, so the result is reproducible).
__global__ void k_normalize(const float* __restrict__ in, int64_t n, int dim, int normalize,
                            _Float16* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t row = wave; row < n; row += nwaves) {
    float s = 0.f;
    for (int d = lane; d < dim; d += 64) {
      const float x = in[row * dim + d];
      s = fmaf(x, x, s);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    // F.normalize semantics: x / max(||x||, 1e-12)
    const float inv = normalize ? 1.0f / fmaxf(sqrtf(s), 1e-12f) : 1.0f;
    for (int d = lane; d < dim; d += 64) out[row * dim + d] = (_Float16)(in[row * dim + d] * inv);
  }
}

static inline int grid_for(int64_t work_items, int block) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  return (int)g;
}

void rf_launch_tile_rows(const void* rows, uint4* tiles, int64_t first_row, int64_t n, int KS,
                         hipStream_t st) {
  const int64_t groups = (n + 31) / 32;
  hipLaunchKernelGGL(k_tile_rows, dim3(grid_for(groups * KS * 2 * 32, 256)), dim3(256), 0, st,
                     (const uint4*)rows, tiles, first_row, n, KS);
}

extern "C" int rf_index_reset(rf_index_t* ix, void* stream) {
  if (!ix) {
    rf_set_error("rf_index_reset: null index");
    return RF_ERR_INVALID;
  }
  ix->size = 0;
  RF_HIP(hipMemsetAsync(ix->max_norm2, 0, 256, (hipStream_t)stream));
  return RF_OK;
}

extern "C" int rf_index_add_f16(rf_index_t* ix, const void* rows_dev, int64_t n, void* stream) {
  if (!ix || (!rows_dev && n > 0)) {
    rf_set_error("rf_index_add_f16: null argument");
    return RF_ERR_INVALID;
  }
  if (n == 0) return RF_OK;
  if (n < 0 || ((uintptr_t)rows_dev & 15) != 0) {
    rf_set_error("rf_index_add_f16: bad n or misaligned rows");
    return RF_ERR_INVALID;
  }
  if (ix->size + n > ix->capacity) {
    rf_set_error("rf_index_add_f16: %lld + %lld rows exceeds capacity %lld", (long long)ix->size,
                 (long long)n, (long long)ix->capacity);
    return RF_ERR_CAPACITY;
  }
  hipStream_t st = (hipStream_t)stream;
  const int KS = ix->KS;
  if (ix->size == 0) RF_HIP(hipMemsetAsync(ix->max_norm2, 0, 256, st));
  // zero every block that is not yet (fully) written: pad rows must read as 0
  const int64_t first_new_block = blocks_for(ix->size);  // first block with no live rows
  const int64_t end_block = blocks_for(ix->size + n);
  if (end_block > first_new_block) {
    const size_t cnt = (size_t)(end_block - first_new_block) * KS * 64;
    hipLaunchKernelGGL(k_zero_u4, dim3(grid_for((int64_t)cnt, 256)), dim3(256), 0, st,
                       ix->tiles + (size_t)first_new_block * KS * 64, cnt);
  }
  rf_launch_tile_rows(rows_dev, ix->tiles, ix->size, n, KS, st);
  hipLaunchKernelGGL(k_max_norm2, dim3(grid_for(n * 64, 256)), dim3(256), 0, st,
                     (const _Float16*)rows_dev, n, ix->dim, ix->max_norm2);
  RF_HIP(hipGetLastError());
  ix->size += n;
  return RF_OK;
}

extern "C" int rf_index_get_rows_f16(const rf_index_t* ix, const int64_t* rows_dev, int64_t n,
                                     void* out_dev, void* stream) {
  if (!ix || !rows_dev || !out_dev || n < 0) {
    rf_set_error("rf_index_get_rows_f16: bad argument");
    return RF_ERR_INVALID;
  }
  if (n == 0) return RF_OK;
  hipLaunchKernelGGL(k_get_rows, dim3(grid_for(n * ix->KS * 2, 256)), dim3(256), 0,
                     (hipStream_t)stream, ix->tiles, rows_dev, n, ix->KS, ix->size,
                     (uint4*)out_dev);
  RF_HIP(hipGetLastError());
  return RF_OK;
}

extern "C" int rf_normalize_f32_to_f16(const float* in_dev, int64_t n, int dim, int normalize,
                                       void* out_dev, void* stream) {
  if (!in_dev || !out_dev || n < 0 || dim <= 0) {
    rf_set_error("rf_normalize_f32_to_f16: bad argument");
    return RF_ERR_INVALID;
  }
  if (n == 0) return RF_OK;
  hipLaunchKernelGGL(k_normalize, dim3(grid_for(n * 64, 256)), dim3(256), 0, (hipStream_t)stream,
                     in_dev, n, dim, normalize, (_Float16*)out_dev);
  RF_HIP(hipGetLastError());
  return RF_OK;
}
