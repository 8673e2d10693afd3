// fetch_calib -- what the TCC fetch counters (rocprofv3 --pmc FETCH_SIZE, TCC_EA0_RDREQ, TCC_EA0_RDREQ_32B) report for the
// access shapes of the path integrator's queues, on KNOWN byte counts.  Measurement tooling: not part of the product library.
//
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- build/fetch_calib <out.json>
//
// Every dispatch reads a known number of records of a known size from a 4 GiB buffer (16 x the Infinity Cache), each record
// exactly once, 16 bytes per load instruction as the product's kernels do, and reports bytes requested and its duration (HIP
// events).  tools/fetch_calibration_summary.py joins that with the counter CSV: counter bytes / requested bytes per shape.
//   stream      : lane i reads record i (coalesced): the shape the guide's "FETCH_SIZE reports half" rule was calibrated on
//   gather      : lane i reads record perm(i), a bijection over ALL slots of the buffer region (neighbours are never co-resident)
//   sparse      : lane i reads record perm(i) of a region 8 x larger than what is read (at most one record per line)
//   pooled      : a wave gathers 64 records scattered over its own window of 320 consecutive records and comes back for the
//                 window's other records in its next four steps (k_bounce's per-category pools over a chunk of the queue)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

enum { STREAM = 0, GATHER = 1, SPARSE = 2, POOLED = 3 };

// a bijection on [0, 2^bits): odd multiplier, xor-shift, odd multiplier
__device__ __forceinline__ uint32_t perm32(uint32_t i, int bits) {
  const uint32_t mask = bits == 32 ? 0xffffffffu : ((1u << bits) - 1u);
  uint32_t x = (i * 0x9E3779B1u) & mask;
  x ^= x >> (bits / 2 + 1);
  x = (x * 0x85EBCA6Bu) & mask;
  return x;
}

// REC: record bytes (multiple of 16); STRIDE: slot pitch in bytes; n: records read; slots_log2: slots in the region
template <int REC, int STRIDE, int PATTERN>
__global__ __launch_bounds__(256) void k_read(const unsigned char* __restrict__ buf, uint32_t n, int slots_log2, double* __restrict__ out) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  uint32_t slot;
  if (PATTERN == STREAM) {
    slot = i;
  } else if (PATTERN == POOLED) {
    // wave w owns windows of 320 records; its step s (0..4) reads 64 of them: record (lane * 5 + s) permuted inside the window
    const uint32_t wave = i >> 6, lane = i & 63u;
    const uint32_t window = wave / 5u, step = wave % 5u;
    const uint32_t j = (lane * 5u + step) * 77u % 320u; /* 77 and 320 are coprime: a bijection inside the window */
    slot = window * 320u + j;
  } else {
    slot = perm32(i, slots_log2);
  }
  const uint4* p = (const uint4*)(buf + (size_t)slot * STRIDE);
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < REC / 16; ++k) {
    const uint4 v = p[k];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[i & 1023u] = 1.0; /* keeps the loads alive; never true on the zero-filled buffer's hash */
}

struct Row { std::string kernel, pattern; int rec, stride; uint32_t n; double requested, ms; };

template <int REC, int STRIDE, int PATTERN>
static Row run(const char* pattern, const unsigned char* buf, size_t buf_bytes, double* out) {
  // records read per dispatch: 2^24 (POOLED: a multiple of 320 * ... waves of 64)
  uint32_t n = 1u << 24;
  int slots_log2 = 24;
  if (PATTERN == SPARSE) slots_log2 = 27;
  while (((size_t)1 << slots_log2) * STRIDE > buf_bytes) { --slots_log2; if (PATTERN != SPARSE) n >>= 1; }
  if (PATTERN == SPARSE && ((size_t)n << 3) > ((size_t)1 << slots_log2)) n = (uint32_t)(((size_t)1 << slots_log2) >> 3);
  if (PATTERN == POOLED) n = (n / (320u * 64u)) * (320u * 64u);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const dim3 grid((n + 255u) / 256u), block(256);
  hipLaunchKernelGGL((k_read<REC, STRIDE, PATTERN>), grid, block, 0, 0, buf, n, slots_log2, out); /* warm-up (also counted) */
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((k_read<REC, STRIDE, PATTERN>), grid, block, 0, 0, buf, n, slots_log2, out);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  char name[96];
  snprintf(name, sizeof name, "k_read<%d, %d, %d>", REC, STRIDE, PATTERN);
  return Row{name, pattern, REC, STRIDE, n, (double)n * REC, ms};
}

int main(int argc, char** argv) {
  const char* out_path = argc > 1 ? argv[1] : "fetch_calib.json";
  const size_t buf_bytes = (size_t)4 << 30;
  unsigned char* buf;
  double* out;
  CHECK(hipMalloc(&buf, buf_bytes));
  CHECK(hipMalloc(&out, 1024 * sizeof(double)));
  CHECK(hipMemset(buf, 0, buf_bytes));
  CHECK(hipDeviceSynchronize());
  std::vector<Row> rows;
  // the queue's shapes: 32-byte path record, 48-byte ray record (16-byte aligned, straddles lines), 64-byte {path, emission}
  // pair, 80-byte triangle slot, 112 bytes = ray + pair read by one lane from two arrays is covered by its parts
  rows.push_back(run<16, 16, STREAM>("stream", buf, buf_bytes, out));
  rows.push_back(run<32, 32, STREAM>("stream", buf, buf_bytes, out));
  rows.push_back(run<48, 48, STREAM>("stream", buf, buf_bytes, out));
  rows.push_back(run<64, 64, STREAM>("stream", buf, buf_bytes, out));
  rows.push_back(run<32, 32, GATHER>("gather", buf, buf_bytes, out));
  rows.push_back(run<48, 48, GATHER>("gather", buf, buf_bytes, out));
  rows.push_back(run<64, 64, GATHER>("gather", buf, buf_bytes, out));
  rows.push_back(run<80, 80, GATHER>("gather", buf, buf_bytes, out));
  rows.push_back(run<16, 16, SPARSE>("sparse", buf, buf_bytes, out));
  rows.push_back(run<32, 32, SPARSE>("sparse", buf, buf_bytes, out));
  rows.push_back(run<32, 64, SPARSE>("sparse", buf, buf_bytes, out));
  rows.push_back(run<48, 48, SPARSE>("sparse", buf, buf_bytes, out));
  rows.push_back(run<64, 64, SPARSE>("sparse", buf, buf_bytes, out));
  rows.push_back(run<80, 80, SPARSE>("sparse", buf, buf_bytes, out));
  rows.push_back(run<32, 32, POOLED>("pooled", buf, buf_bytes, out));
  rows.push_back(run<48, 48, POOLED>("pooled", buf, buf_bytes, out));
  rows.push_back(run<64, 64, POOLED>("pooled", buf, buf_bytes, out));
  FILE* f = fopen(out_path, "w");
  if (!f) { perror(out_path); return 1; }
  fprintf(f, "{\"buffer_bytes\": %zu, \"dispatches_per_row\": 2, \"rows\": [\n", buf_bytes);
  for (size_t i = 0; i < rows.size(); ++i) {
    const Row& r = rows[i];
    fprintf(f, "  {\"kernel\": \"%s\", \"pattern\": \"%s\", \"record_bytes\": %d, \"stride\": %d, \"records\": %u, \"requested_bytes\": %.0f, \"ms\": %.4f, \"requested_gbs\": %.1f}%s\n",
            r.kernel.c_str(), r.pattern.c_str(), r.rec, r.stride, r.n, r.requested, r.ms, r.requested / (r.ms * 1e-3) * 1e-9, i + 1 < rows.size() ? "," : "");
    printf("%-22s %-7s rec %3d stride %3d: %8.1f MB requested, %7.3f ms, %7.1f GB/s requested\n", r.kernel.c_str(), r.pattern.c_str(), r.rec, r.stride, r.requested * 1e-6, r.ms, r.requested / (r.ms * 1e-3) * 1e-9);
  }
  fprintf(f, "]}\n");
  fclose(f);
  CHECK(hipFree(buf));
  CHECK(hipFree(out));
  return 0;
}
