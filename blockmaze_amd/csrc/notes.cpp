// Host-side note hashing, hex blobs and the commitment Merkle tree (rows W2 of SURVEY.md §8a).
// Sources: libsnark-vnt/src/send/uint256.h:190-248 (GetHex / SetHex), send/Note.h:30-78, send/util.h:18-26,233-258,
// deposit/IncrementalMerkleTree.tcc:14-24 (SHA256Compress::combine = one compression of left||right from the standard IV,
// no padding), deposit/depositcgo.cpp:302-325 (genRoot).  SHA-256 itself is FIPS 180-4.
#include <cctype>
#include <cstring>
#include "blockmaze_circuits.hpp"

namespace zk {

static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74,
        0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3,
        0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819,
        0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa,
        0xa4506ceb, 0xbef9a3f7, 0xc67178f2
  };
static const uint32_t H256[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
static inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static void compress(uint32_t s[8], const uint8_t blk[64]) {
  uint32_t w[64];
  for (int i = 0; i < 16; i++) w[i] = (uint32_t)blk[4 * i] << 24 | (uint32_t)blk[4 * i + 1] << 16 | (uint32_t)blk[4 * i + 2] << 8 | blk[4 * i + 3];
  for (int i = 16; i < 64; i++) {
    uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  uint32_t a = s[0], b = s[1], c = s[2], d = s[3], e = s[4], f = s[5], g = s[6], h = s[7];
  for (int i = 0; i < 64; i++) {
    uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g), t1 = h + S1 + ch + K256[i] + w[i], S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a,
        22), mj = (a & b) ^ (a & c) ^ (b & c), t2 = S0 + mj;
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2; }
  s[0] += a; s[1] += b; s[2] += c; s[3] += d; s[4] += e; s[5] += f; s[6] += g; s[7] += h; }
static void state_out(const uint32_t s[8], uint8_t out[32]) {
  for (int i = 0; i < 8; i++) {
    out[4 * i] = s[i] >> 24;
    out[4 * i + 1] = s[i] >> 16;
    out[4 * i + 2] = s[i] >> 8;
    out[4 * i + 3] = s[i];
  }
}
void sha256(const uint8_t *msg, size_t len, uint8_t out[32]) {
  uint32_t s[8]; memcpy(s, H256, 32); size_t off = 0; for (; off + 64 <= len; off += 64) compress(s, msg + off);
  uint8_t tail[128];
  size_t rem = len - off;
  memset(tail, 0, sizeof tail);
  memcpy(tail, msg + off, rem);
  tail[rem] = 0x80;
  size_t tl = rem + 9 <= 64 ? 64 : 128;
  uint64_t bits = (uint64_t)len * 8;
  for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i)); compress(s, tail); if (tl == 128) compress(s, tail + 64); state_out(s, out); }
void sha256_compress_raw(const uint8_t left[32], const uint8_t right[32], uint8_t out[32]) {
  uint32_t s[8];
  memcpy(s, H256, 32);
  uint8_t blk[64];
  memcpy(blk, left, 32);
  memcpy(blk + 32, right, 32);
  compress(s, blk);
  state_out(s, out);
}

static int hexdigit(char c) {
  if (c >= '0' && c <= '9') return c - '0';
  if (c >= 'a' && c <= 'f') return c - 'a' + 10;
  if (c >= 'A' && c <= 'F') return c - 'A' + 10;
  return -1;
}
static void set_hex(uint8_t *data, size_t width, const char *psz) {   // uint256.h:222-248
  memset(data, 0, width); if (!psz) return; while (isspace((unsigned char)*psz)) psz++; if (psz[0] == '0' && tolower((unsigned char)psz[1]) == 'x') psz += 2;
  const char *pbegin = psz; while (hexdigit(*psz) != -1) psz++; psz--; uint8_t *p1 = data, *pend = data + width;
  while (psz >= pbegin && p1 < pend) { *p1 = (uint8_t)hexdigit(*psz--); if (psz >= pbegin) { *p1 |= (uint8_t)(hexdigit(*psz--) << 4); p1++; } } }
Blob256 blob256_from_hex(const char *s) { Blob256 r; set_hex(r.b, 32, s); return r; }
Blob160 blob160_from_hex(const char *s) { Blob160 r; set_hex(r.b, 20, s); return r; }
std::string blob_to_hex(const uint8_t *b, size_t n) {
  static const char *d = "0123456789abcdef";
  std::string s(2 * n, '0');
  for (size_t i = 0; i < n; i++) {
    s[2 * i] = d[b[n - 1 - i] >> 4];
    s[2 * i + 1] = d[b[n - 1 - i] & 15];
  }
  return s;
}

static void le64(uint64_t v, uint8_t o[8]) { for (int i = 0; i < 8; i++) o[i] = (uint8_t)(v >> (8 * i)); }
Blob256 note_cm(uint64_t value, const Blob256 &sn, const Blob256 &r) {
  uint8_t m[72];
  le64(value, m);
  memcpy(m + 8, sn.b, 32);
  memcpy(m + 40, r.b, 32);
  Blob256 o;
  sha256(m, 72, o.b);
  return o;
}
Blob256 note_s_cm(uint64_t value, const Blob160 &pk, const Blob256 &r, const Blob256 &sn_old) {
  uint8_t m[92];
  le64(value, m);
  memcpy(m + 8, pk.b, 20);
  memcpy(m + 28, r.b, 32);
  memcpy(m + 60, sn_old.b, 32);
  Blob256 o;
  sha256(m, 92, o.b);
  return o;
}
Blob256 compute_prf(const Blob256 &sk, const Blob256 &r) {
  uint8_t m[64];
  memcpy(m, sk.b, 32);
  memcpy(m + 32, r.b, 32);
  Blob256 o;
  sha256(m, 64, o.b);
  return o;
}
Blob256 compute_crh(const Blob160 &pk, const Blob256 &r) {
  uint8_t m[52];
  memcpy(m, pk.b, 20);
  memcpy(m + 20, r.b, 32);
  Blob256 o;
  sha256(m, 52, o.b);
  return o;
}

// Full tree over 2^depth leaves with uncommitted (all-zero) leaves after the appended ones: the root of the incremental tree
// after n appends (IncrementalMerkleTree.tcc:179-258 fills the unseen part with empty roots, which is the same tree).
static std::vector<std::vector<Blob256>> tree_levels(const std::vector<Blob256> &leaves, size_t depth) {
  std::vector<Blob256> empty(depth + 1);
  memset(empty[0].b, 0, 32);
  for (size_t d = 1; d <= depth; d++) sha256_compress_raw(empty[d - 1].b, empty[d - 1].b, empty[d].b);
  std::vector<std::vector<Blob256>> lv(depth + 1); lv[0] = leaves;
  for (size_t d = 0; d < depth; d++) { const auto &cur = lv[d]; auto &up = lv[d + 1]; up.resize((cur.size() + 1) / 2);
    for (size_t i = 0; i < up.size(); i++) {
      const Blob256 &l = cur[2 * i], &r = 2 * i + 1 < cur.size() ? cur[2 * i + 1] : empty[d];
      sha256_compress_raw(l.b, r.b, up[i].b);
    }
  }
  if (lv[depth].empty()) lv[depth].push_back(empty[depth]);
  lv.push_back(empty);   // stash the empty roots as an extra entry
  return lv; }
Blob256 merkle_root(const std::vector<Blob256> &leaves, size_t depth) { return tree_levels(leaves, depth)[depth][0]; }
std::vector<Blob256> merkle_path(const std::vector<Blob256> &leaves, size_t depth, size_t index, std::vector<bool> &index_bits) {
  auto lv = tree_levels(leaves, depth); const std::vector<Blob256> &empty = lv[depth + 1]; std::vector<Blob256> path(depth); index_bits.assign(depth, false);
  for (size_t d = 0; d < depth; d++) { size_t pos = index >> d, sib = pos ^ 1; path[d] = sib < lv[d].size() ? lv[d][sib] : empty[d]; index_bits[d] = pos & 1; }
  return path; }   // path[0] = sibling at the leaf level

}  // namespace zk
