/* The C ABI from plain C: include/vita_hip.h must compile as C (no C++ / torch types), every entry point must resolve,
 * and argument validation must answer without touching a GPU.  Built and run by tests/test_cpu_host.py. */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "vita_hip.h"

#define CHECK(cond)                                                   \
  do {                                                                \
    if (!(cond)) { printf("FAILED: %s (line %d)\n", #cond, __LINE__); return 1; } \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) { printf("dlopen: %s\n", dlerror()); return 3; }
  int (*abi)(void) = (int (*)(void))dlsym(h, "vita_abi_version");
  const char* (*errstr)(int) = (const char* (*)(int))dlsym(h, "vita_error_string");
  int (*rms)(const void*, const void*, void*, float*, int64_t, int, float, void*) =
      (int (*)(const void*, const void*, void*, float*, int64_t, int, float, void*))dlsym(h, "vita_rmsnorm_fwd");
  int (*gemm)(const void*, int64_t, const void*, int64_t, void*, int64_t, int64_t, int64_t, int64_t, int, const void*,
              const void*, const void*, int64_t, void*) =
      (int (*)(const void*, int64_t, const void*, int64_t, void*, int64_t, int64_t, int64_t, int64_t, int, const void*,
               const void*, const void*, int64_t, void*))dlsym(h, "vita_gemm_bf16");
  int (*attn)(const vita_attn_params*, void*) = (int (*)(const vita_attn_params*, void*))dlsym(h, "vita_flash_attn_fwd");
  int (*dlayer)(const vita_decode_layer_params*, void*) =
      (int (*)(const vita_decode_layer_params*, void*))dlsym(h, "vita_decode_layer_attn");
  CHECK(abi && errstr && rms && gemm && attn && dlayer);
  CHECK(abi() >= 10);
  CHECK(strcmp(errstr(VITA_OK), "ok") == 0);
  CHECK(rms(NULL, NULL, NULL, NULL, 4, 64, 1e-6f, NULL) == VITA_ERR_INVALID_ARG);
  char dummy[16];
  CHECK(gemm(dummy, 64, dummy, 64, dummy, 64, 4, 64, 63, VITA_EPI_NONE, NULL, NULL, NULL, 0, NULL) == VITA_ERR_UNSUPPORTED);
  CHECK(gemm(dummy, 64, dummy, 64, dummy, 64, 0, 64, 64, VITA_EPI_NONE, NULL, NULL, NULL, 0, NULL) == VITA_OK); /* M = 0 */
  vita_attn_params ap;
  memset(&ap, 0, sizeof ap);
  CHECK(attn(&ap, NULL) == VITA_ERR_INVALID_ARG);
  CHECK(attn(NULL, NULL) == VITA_ERR_INVALID_ARG);
  vita_decode_layer_params dp;
  memset(&dp, 0, sizeof dp);
  CHECK(dlayer(&dp, NULL) == VITA_ERR_INVALID_ARG);
  printf("C ABI OK: version %d, sizeof(vita_attn_params) = %zu, sizeof(vita_decode_layer_params) = %zu\n", abi(),
         sizeof(vita_attn_params), sizeof(vita_decode_layer_params));
  return 0;
}
