#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "tahoe_amd.h"
int main(int argc, char** argv) {
  const char* dir = argv[1];
  for (const char* name : {"susy_like_c18", "k3_like_c256", "depth0", "depth1_ties"}) {
    std::string m = std::string(dir) + "/" + name + ".model.txt", d = std::string(dir) + "/" + name + ".data.txt";
    for (const char* th : {"1", "3", "8"}) {
      setenv("TAHOE_LOADER_THREADS", th, 1);
      int T = 10, D = 20, R = 1000, C = 500; float miss = 0; tahoe_dense_node* n = nullptr; float* x = nullptr;
      if (tahoe_load_model(m.c_str(), &T, &D, &n) || tahoe_load_data(d.c_str(), &R, &C, &miss, &x)) { printf("load failed %s\n", tahoe_last_error()); return 1; }
      std::string mb = std::string("/tmp/tahoe_asan/") + name + ".m.tbin", db = std::string("/tmp/tahoe_asan/") + name + ".d.tbin";
      if (tahoe_save_model_bin(mb.c_str(), T, D, n) || tahoe_save_data_bin(db.c_str(), R, C, miss, x)) { printf("save failed\n"); return 1; }
      int T2, D2, R2, C2; float m2; tahoe_dense_node* n2 = nullptr; float* x2 = nullptr;
      if (tahoe_load_model_bin(mb.c_str(), &T2, &D2, &n2) || tahoe_load_data_bin(db.c_str(), &R2, &C2, &m2, &x2)) { printf("bin load failed\n"); return 1; }
      size_t nn = (size_t)T * tahoe_tree_num_nodes(D);
      if (T2 != T || D2 != D || memcmp(n, n2, nn * sizeof(*n)) || memcmp(x, x2, (size_t)R * C * 4)) { printf("mismatch\n"); return 1; }
      tahoe_free_host(n); tahoe_free_host(x); tahoe_free_host(n2); tahoe_free_host(x2);
    }
  }
  // truncated / empty files
  FILE* f = fopen("/tmp/tahoe_asan/short.txt", "w"); fputs("2\n3\n-999\n0.5\n1.5", f); fclose(f);
  int R = 1, C = 1; float miss = 0; float* x = nullptr;
  if (tahoe_load_data("/tmp/tahoe_asan/short.txt", &R, &C, &miss, &x)) return 1;
  printf("short: %d x %d last %g\n", R, C, x[R * C - 1]); tahoe_free_host(x);
  f = fopen("/tmp/tahoe_asan/empty.txt", "w"); fclose(f);
  R = 2; C = 2; if (tahoe_load_data("/tmp/tahoe_asan/empty.txt", &R, &C, &miss, &x)) return 1; tahoe_free_host(x);
  printf("asan driver ok\n");
  return 0;
}
