// muavta_rng.h — MT19937 block primitives of CPython's random.Random on the device (included by muavta_device.h inside namespace muavta).
// ====================================================================================================
// CPython random.Random on a per-env tape in HBM: each stream keeps two consecutive raw MT19937
// blocks (2 x 624 words); lane 0 tempers words at the cursor, and the whole wave regenerates a
// consumed block at a step boundary (Modules/_randommodule.c genrand_uint32 / init_by_array).
// ====================================================================================================
enum { ST_AGENT = 0, ST_OBS = 1, ST_TGT = 2, ST_MISSION = 3 };

DEV uint32_t mt_mix(uint32_t a, uint32_t b, uint32_t m) {
  uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return m ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
// dst = next MT block after src (src/dst: LDS or global, distinct buffers).  All lanes.
DEV void mt_twist(const uint32_t* src, uint32_t* dst) {
  const int lane = threadIdx.x;
  for (int k = lane; k < 227; k += WG) dst[k] = mt_mix(src[k], src[k + 1], src[k + 397]);
  __syncthreads();
  for (int k = 227 + lane; k < 454; k += WG) dst[k] = mt_mix(src[k], src[k + 1], dst[k - 227]);
  __syncthreads();
  for (int k = 454 + lane; k < 623; k += WG) dst[k] = mt_mix(src[k], src[k + 1], dst[k - 227]);
  __syncthreads();
  if (lane == 0) dst[623] = mt_mix(src[623], dst[0], dst[396]);
  __syncthreads();
}
// Same on LDS buffers of a single-wave workgroup: phases are ordered by lds_sync() (no vmcnt drain).
DEV void mt_twist_lds(const uint32_t* src, uint32_t* dst) {
  const int lane = threadIdx.x;
  for (int k = lane; k < 227; k += WG) dst[k] = mt_mix(src[k], src[k + 1], src[k + 397]);
  lds_sync();
  for (int k = 227 + lane; k < 454; k += WG) dst[k] = mt_mix(src[k], src[k + 1], dst[k - 227]);
  lds_sync();
  for (int k = 454 + lane; k < 623; k += WG) dst[k] = mt_mix(src[k], src[k + 1], dst[k - 227]);
  lds_sync();
  if (lane == 0) dst[623] = mt_mix(src[623], dst[0], dst[396]);
  lds_sync();
}
// init_by_array(key[0..len)) into mt[624] (LDS).  One lane.
DEV void mt_seed(uint32_t* mt, uint32_t k0, uint32_t k1, int len) {
  uint32_t g = 19650218u;  // init_genrand(19650218) generated on the fly
  uint32_t prev = g;
  mt[0] = g;
  int j = 0;
  for (int i = 1; i < 624; i++) {
    g = 1812433253u * (g ^ (g >> 30)) + (uint32_t)i;
    uint32_t key = j ? k1 : k0;
    prev = (g ^ ((prev ^ (prev >> 30)) * 1664525u)) + key + (uint32_t)j;
    mt[i] = prev;
    j++;
    if (j >= len) j = 0;
  }
  // 624th iteration of the first loop: i wrapped to 1 with mt[0] = mt[623]
  mt[0] = prev;
  {
    uint32_t key = j ? k1 : k0;
    prev = (mt[1] ^ ((prev ^ (prev >> 30)) * 1664525u)) + key + (uint32_t)j;
    mt[1] = prev;
  }
  // second loop (i = 2..623): the recurrence is serial in `prev`, but the mt[i] operands are first-loop values
  // whose addresses are known, so they are fetched eight at a time (one LDS wait per 8 steps, not per step)
  {
    int i = 2;
    for (; i + 8 <= 624; i += 8) {
      uint32_t m[8];
#pragma unroll
      for (int q = 0; q < 8; q++) m[q] = mt[i + q];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        prev = (m[q] ^ ((prev ^ (prev >> 30)) * 1566083941u)) - (uint32_t)(i + q);
        m[q] = prev;
      }
#pragma unroll
      for (int q = 0; q < 8; q++) mt[i + q] = m[q];
    }
    for (; i < 624; i++) {
      prev = (mt[i] ^ ((prev ^ (prev >> 30)) * 1566083941u)) - (uint32_t)i;
      mt[i] = prev;
    }
  }
  mt[0] = prev;
  prev = (mt[1] ^ ((prev ^ (prev >> 30)) * 1566083941u)) - 1u;
  mt[1] = prev;
  mt[0] = 0x80000000u;
}
