// hostreg_bench.hip — two questions behind the CLI's host stage:
//  1. what does a HIP process pay before its first kernel can run (runtime bring-up, first pinned buffer, first
//     device allocation, code-object load)?  The floor under `first_submit_at_s`.
//  2. can a file in the page cache be handed to the GPU without a CPU copy?  mmap + hipHostRegister + H2D straight
//     from the mapping, against pread() into a pinned buffer + H2D.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/hostreg_bench tools/hostreg_bench.hip
//   ./tools/hostreg_bench [file in /dev/shm, >= 1 GiB; created when missing]
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <thread>
#include <vector>

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void k_touch(const unsigned *p, unsigned *out, size_t n) {
  unsigned acc = 0;
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
  if (acc == 0x12345678u) *out = acc;
}

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                               \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

int main(int argc, char **argv) {
  const double t0 = now_s();
  int n_dev = 0;
  CK(hipGetDeviceCount(&n_dev));
  const double t1 = now_s();
  CK(hipSetDevice(0));
  CK(hipFree(nullptr));
  const double t2 = now_s();
  const size_t chunk = 64u << 20;
  void *pin0 = nullptr, *pin1 = nullptr, *dev = nullptr;
  CK(hipHostMalloc(&pin0, chunk, hipHostMallocPortable));
  const double t3 = now_s();
  CK(hipHostMalloc(&pin1, chunk, hipHostMallocPortable));
  const double t4 = now_s();
  CK(hipMalloc(&dev, chunk));
  const double t5 = now_s();
  unsigned *d_out = nullptr;
  CK(hipMalloc(&d_out, 4));
  hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, 0, (const unsigned *)dev, d_out, chunk / 4);
  CK(hipDeviceSynchronize());
  const double t6 = now_s();
  printf("startup: hipGetDeviceCount %.3f s, hipSetDevice+hipFree(0) %.3f, first hipHostMalloc(64 MiB) %.3f, second %.3f, "
         "hipMalloc(64 MiB) %.3f, first kernel (code object load) %.3f; total %.3f s\n",
         t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t6 - t0);

  const char *path = argc > 1 ? argv[1] : "/dev/shm/hostreg_bench.bin";
  const size_t total = 2ull << 30;
  struct stat st;
  bool made = false;
  if (stat(path, &st) != 0 || (size_t)st.st_size < total) {
    int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return 1;
    std::vector<char> buf(chunk);
    for (size_t i = 0; i < chunk; i++) buf[i] = (char)(i * 2654435761u >> 13);
    for (size_t off = 0; off < total; off += chunk)
      if (write(fd, buf.data(), chunk) != (ssize_t)chunk) return 1;
    close(fd);
    made = true;
  }
  int fd = open(path, O_RDONLY);
  if (fd < 0) return 1;
  hipStream_t st0;
  CK(hipStreamCreateWithFlags(&st0, hipStreamNonBlocking));

  // ---- A: pread into a pinned buffer (1, 8 threads), then H2D
  for (int n_thr : {1, 8}) {
    double t_read = 0, t_h2d = 0;
    for (size_t off = 0; off < total; off += chunk) {
      const double a = now_s();
      std::vector<std::thread> th;
      for (int t = 0; t < n_thr; t++)
        th.emplace_back([&, t]() {
          const size_t lo = chunk * t / n_thr, hi = chunk * (t + 1) / n_thr;
          size_t done = 0;
          while (lo + done < hi) {
            ssize_t g = pread(fd, (char *)pin0 + lo + done, hi - lo - done, (off_t)(off + lo + done));
            if (g <= 0) break;
            done += (size_t)g;
          }
        });
      for (auto &x : th) x.join();
      const double b = now_s();
      CK(hipMemcpyAsync(dev, pin0, chunk, hipMemcpyHostToDevice, st0));
      CK(hipStreamSynchronize(st0));
      t_read += b - a;
      t_h2d += now_s() - b;
    }
    printf("pread x%d -> pinned -> H2D: read %.1f GB/s, H2D %.1f GB/s (serial sum %.1f GB/s)\n", n_thr, total / t_read / 1e9,
           total / t_h2d / 1e9, total / (t_read + t_h2d) / 1e9);
  }

  // ---- B: mmap the file, register 64 MiB pieces, H2D straight from the page cache
  for (int populate = 0; populate < 2; populate++) {
    const double m0 = now_s();
    void *map = mmap(nullptr, total, PROT_READ, MAP_SHARED | (populate ? MAP_POPULATE : 0), fd, 0);
    if (map == MAP_FAILED) {
      printf("mmap failed\n");
      return 1;
    }
    const double m1 = now_s();
    double t_reg = 0, t_h2d = 0, t_unreg = 0;
    bool ok = true;
    for (size_t off = 0; off < total && ok; off += chunk) {
      const double a = now_s();
      hipError_t e = hipHostRegister((char *)map + off, chunk, hipHostRegisterDefault);
      if (e != hipSuccess) {
        printf("hipHostRegister(mmap of %s): %s\n", path, hipGetErrorString(e));
        (void)hipGetLastError();
        ok = false;
        break;
      }
      const double b = now_s();
      CK(hipMemcpyAsync(dev, (char *)map + off, chunk, hipMemcpyHostToDevice, st0));
      CK(hipStreamSynchronize(st0));
      const double c = now_s();
      CK(hipHostUnregister((char *)map + off));
      t_reg += b - a;
      t_h2d += c - b;
      t_unreg += now_s() - c;
    }
    if (ok)
      printf("mmap%s (%.3f s) + hipHostRegister per 64 MiB: register %.1f GB/s, H2D %.1f GB/s, unregister %.1f GB/s (serial sum %.1f GB/s)\n",
             populate ? " MAP_POPULATE" : "", m1 - m0, total / t_reg / 1e9, total / t_h2d / 1e9, total / t_unreg / 1e9,
             total / (t_reg + t_h2d + t_unreg) / 1e9);
    // ---- C: H2D from the mapping without registering it (the runtime stages pageable memory itself)
    {
      const double a = now_s();
      for (size_t off = 0; off < total; off += chunk) {
        CK(hipMemcpyAsync(dev, (char *)map + off, chunk, hipMemcpyHostToDevice, st0));
        CK(hipStreamSynchronize(st0));
      }
      printf("mmap%s, H2D from the unregistered mapping: %.1f GB/s\n", populate ? " MAP_POPULATE" : "", total / (now_s() - a) / 1e9);
    }
    munmap(map, total);
  }
  // ---- D: register the whole mapping once
  {
    void *map = mmap(nullptr, total, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, 0);
    const double a = now_s();
    hipError_t e = hipHostRegister(map, total, hipHostRegisterDefault);
    const double b = now_s();
    if (e == hipSuccess) {
      double t_h2d = 0;
      for (size_t off = 0; off < total; off += chunk) {
        const double c = now_s();
        CK(hipMemcpyAsync(dev, (char *)map + off, chunk, hipMemcpyHostToDevice, st0));
        CK(hipStreamSynchronize(st0));
        t_h2d += now_s() - c;
      }
      printf("register the whole 2 GiB mapping once: %.3f s (%.1f GB/s), H2D %.1f GB/s\n", b - a, total / (b - a) / 1e9, total / t_h2d / 1e9);
      CK(hipHostUnregister(map));
    } else {
      printf("hipHostRegister(whole mapping): %s\n", hipGetErrorString(e));
    }
    munmap(map, total);
  }
  close(fd);
  if (made) unlink(path);
  return 0;
}
