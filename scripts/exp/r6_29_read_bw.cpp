#include <fcntl.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <thread>
#include <vector>
int main(int argc, char **argv) {
  const char *path = argv[1];
  const size_t total = (size_t)atol(argv[2]) << 20, blk = 100u << 20;
  int nt = atoi(argv[3]);
  std::vector<char> buf(blk);
  for (size_t i = 0; i < blk; i += 4096) buf[i] = 1;
  auto t0 = std::chrono::steady_clock::now();
  if (nt == 0) {
    std::ifstream f(path, std::ios::binary);
    for (size_t o = 0; o + blk <= total; o += blk) { f.seekg(o); f.read(buf.data(), blk); }
  } else {
    int fd = open(path, O_RDONLY);
    for (size_t o = 0; o + blk <= total; o += blk) {
      std::vector<std::thread> th;
      const size_t per = (blk / nt + 4095) & ~(size_t)4095;
      for (int k = 0; k < nt; ++k) th.emplace_back([&, k] {
        size_t b = k * per, e = std::min(blk, b + per);
        while (b < e) { ssize_t r = pread(fd, buf.data() + b, e - b, o + b); if (r <= 0) break; b += r; }
      });
      for (auto &t : th) t.join();
    }
    close(fd);
  }
  double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("threads %d: %.2f GB/s\n", nt, total / s / 1e9);
}
