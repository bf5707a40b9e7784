// GLV constants of alt_bn128 (generated; see msm.cuh): phi(x, y) = (beta x, y) = lambda (x, y);  k = k1 + k2 lambda (mod r) with
// c1 = floor(k G1 / 2^256), c2 = floor(k G2 / 2^256), k1 = k - c1 a1 - c2 a2, k2 = c1 |b1| - c2 b2  (lattice basis (a1, b1), (a2, b2) of {(x, y): x + y lambda = 0 mod r}, b1 < 0)
constexpr uint32_t GLV_BETA_MONT[8] = {0xd782e155u, 0x71930c11u, 0xffbe3323u, 0xa6bb947cu, 0xd4741444u, 0xaa303344u, 0x26594943u, 0x2c3b3f0du};   // beta = 0x59e26bcea0d48bacd4f263f1acdb5c4f5763473177fffffe, Montgomery form
constexpr uint32_t GLV_G1[3] = {0xc7e0b3d7u, 0xd91d232eu, 0x00000002u};
constexpr uint32_t GLV_G2[5] = {0x391eb18eu, 0x7a7bd9d4u, 0xa773d2cfu, 0x4ccef014u, 0x00000002u};
constexpr uint32_t GLV_A1[2] = {0x94d213e3u, 0x89d32568u};
constexpr uint32_t GLV_A2[4] = {0x1221250bu, 0x0be4e154u, 0xeeb859fdu, 0x6f4d8248u};
constexpr uint32_t GLV_NB1[4] = {0x7d4f1128u, 0x8211bbebu, 0xeeb859fcu, 0x6f4d8248u};   // -b1
constexpr uint32_t GLV_B2[2] = {0x94d213e3u, 0x89d32568u};
