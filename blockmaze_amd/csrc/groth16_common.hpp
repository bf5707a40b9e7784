// Shared by the four translation units the Groth16 host code is cut into (round 6: groth16.cpp was one unit of 1,900 lines) - groth16_keyio.cpp (key files and the
// fast container), groth16_keygen.cpp (toxic waste, domains, the generator), groth16_prover.cpp (the prover pipeline), groth16_verifier.cpp (verifier, proof encoding,
// the schedule's host interpreters): conversions between the raw records and the host field / curve types, and the few functions one unit borrows from another.
#pragma once
#include <chrono>
#include <cstring>
#include "groth16.hpp"

namespace zk {
using namespace host;


static inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static inline HFq fq_of(const Fe32 &f) { HFq r; memcpy(r.l, &f, 32); return r; }
static inline HFr fr_of(const Fe32 &f) { HFr r; memcpy(r.l, &f, 32); return r; }
static inline Fe32 fe_of(const HFq &f) { Fe32 r; memcpy(&r, f.l, 32); return r; }
static inline Fe32 fe_of_r(const HFr &f) { Fe32 r; memcpy(&r, f.l, 32); return r; }
static inline HG1 g1_of(const G1AffineRaw &p) { return HG1::from_affine(fq_of(p.x), fq_of(p.y)); }
static inline HFq2 fq2_of(const Fe32 &a, const Fe32 &b) { return {fq_of(a), fq_of(b)}; }
static inline HG2 g2_of(const G2AffineRaw &p) { return HG2::from_affine(fq2_of(p.x0, p.x1), fq2_of(p.y0, p.y1)); }
static inline G1AffineRaw raw_of(const HG1 &p) { HFq x, y; p.to_affine(x, y); return {fe_of(x), fe_of(y)}; }
static inline G2AffineRaw raw_of(const HG2 &p) { HFq2 x, y; p.to_affine(x, y); return {fe_of(x.c0), fe_of(x.c1), fe_of(y.c0), fe_of(y.c1)}; }
static inline bool is_zero_raw(const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; for (size_t i = 0; i < n; i++) if (b[i]) return false; return true; }

size_t domain_size_for(size_t min_size);     // groth16_keygen.cpp: the size of the evaluation domain libfqfft picks for min_size points
HG2 default_g2_generator();                   // groth16_keygen.cpp: the generator of G2 the key generator and the default proof use
HFr random_fr();                              // groth16_keygen.cpp: a uniform field element from getrandom(2), Montgomery form
}  // namespace zk
