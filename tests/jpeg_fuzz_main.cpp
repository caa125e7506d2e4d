// Fuzz harness for the host JPEG stage (csrc/jpeg_host.cpp), built by tests/test_jpeg_cpu.py with
// -fsanitize=address,undefined: every file named on the command line is parsed and, when the header is accepted, entropy-decoded
// into a buffer of exactly the size the header asks for.  Any out-of-bounds access aborts the process.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "klab_mm.h"

int main(int argc, char** argv) {
  int accepted = 0, decoded = 0;
  for (int i = 1; i < argc; ++i) {
    FILE* f = fopen(argv[i], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> d((size_t)n);  // exact size: reads past the end are caught
    if (n && fread(d.data(), 1, (size_t)n, f) != (size_t)n) return 2;
    fclose(f);
    klab_jpeg_info info;
    if (klab_jpeg_read_info(d.data(), d.size(), &info) != KLAB_OK) continue;
    ++accepted;
    if (!info.supported || info.coef_blocks <= 0 || info.coef_blocks > (1 << 22)) continue;
    std::vector<short> coefs((size_t)info.coef_blocks * 64);
    unsigned short qt[192];
    klab_jpeg_info info2;
    if (klab_jpeg_entropy_decode(d.data(), d.size(), coefs.data(), qt, &info2) == KLAB_OK) ++decoded;
  }
  printf("files %d accepted %d decoded %d\n", argc - 1, accepted, decoded);
  return 0;
}
