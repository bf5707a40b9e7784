/* mint_key / send_key / deposit_key / redeem_key — the key-generation executables of the reference
 * (libsnark-vnt/src/{mint,send,deposit,redeem}/getpvk.cpp:41-51: build the circuit, run r1cs_gg_ppzksnark_generator, write Xpk.txt and Xvk.txt into the current
 * directory), on the GPU generator of libzkgpu.so: the fixed-base exponentiations of the four queries run on the device (0.8 s for send instead of 41-57 s), the
 * files are in the reference's on-disk format and load in libsnark and in this engine alike.
 *
 * One binary, four names: the circuit is taken from the name it is invoked by (the Makefile links it as X_key), or from the first argument.
 *   send_key                     ->  ./sendpk.txt ./sendvk.txt   (fresh toxic waste from the kernel's CSPRNG, like the reference's std::random_device)
 *   zk_key deposit [out_dir]     ->  <out_dir>/depositpk.txt, depositvk.txt
 * Environment: ZK_KEY_SEED=<n> makes the toxic waste reproducible — TEST KEYS ONLY; ZK_TREE_DEPTH overrides the deposit circuit's Merkle depth (reference: 8). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../../include/zkgpu.h"

int main(int argc, char **argv) {
  static const char *names[4] = {"mint", "send", "deposit", "redeem"}; const char *base = strrchr(argv[0], '/'); base = base ? base + 1 : argv[0]; int kind = -1; const char *dir = ".";
  for (int k = 0; k < 4; k++) if (!strncmp(base, names[k], strlen(names[k]))) kind = k;
  int a = 1; if (kind < 0 && argc > 1) { for (int k = 0; k < 4; k++) if (!strcmp(argv[1], names[k])) kind = k; a = 2; }
  if (kind < 0) { fprintf(stderr, "usage: {mint,send,deposit,redeem}_key [out_dir]   |   zk_key <mint|send|deposit|redeem> [out_dir]\n"); return 2; }
  if (argc > a) dir = argv[a];
  const char *seed_s = getenv("ZK_KEY_SEED"), *depth_s = getenv("ZK_TREE_DEPTH"); unsigned long long seed = seed_s ? strtoull(seed_s, 0, 0) : 0; int depth = depth_s ? atoi(depth_s) : 8;
  char pk[4096], vk[4096]; snprintf(pk, sizeof pk, "%s/%spk.txt", dir, names[kind]); snprintf(vk, sizeof vk, "%s/%svk.txt", dir, names[kind]);
  if (seed) fprintf(stderr, "%s_key: ZK_KEY_SEED is set - these are TEST keys (reproducible toxic waste)\n", names[kind]);
  int rc = zkgpu_keygen(kind, depth, seed, pk, vk);
  if (rc != ZKGPU_OK) { fprintf(stderr, "%s_key: %s\n", names[kind], zkgpu_last_error()); return 1; }
  printf("%s\n%s\n", pk, vk); return 0;
}
