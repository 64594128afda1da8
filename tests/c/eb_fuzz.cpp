// Sanitizer fuzz driver of the host-side C++ behind the C ABI (csrc/easybytes.cpp: the one parser of untrusted wire bytes; csrc/comm.cpp:
// argument checks with no communicator).  Built by `make -C ddrl4nav_amd/csrc asan` with g++ -fsanitize=address,undefined
// -fno-sanitize-recover=all and run by tests/test_asan.py in a child process: any out-of-bounds access or undefined arithmetic aborts.
//   eb_fuzz <seed> <iterations>
// Three legs per iteration: (1) structured round trip (encode random arrays -> scan -> payload bytes equal), (2) every strict prefix class
// (truncation) and random byte mutations of a valid stream through every scanner, (3) hostile headers (32-bit dims whose product passes
// INT64_MAX, near-2^63 lengths).  Wire format: USTC_lab/data/easybytes.py:18-75,141-162.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/ddrl.h"

static uint64_t state;
static uint64_t rnd() {  // splitmix64
  uint64_t z = (state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static void put32(std::vector<uint8_t>& v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s)); }
static void put64(std::vector<uint8_t>& v, uint64_t x) { put32(v, (uint32_t)(x >> 32)); put32(v, (uint32_t)x); }

#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "eb_fuzz: %s failed at line %d (seed state %llu)\n", #c, __LINE__, (unsigned long long)state); std::exit(3); } } while (0)

// every scanner over an arbitrary byte string, into exactly-sized heap buffers (ASan sees any overrun)
static void scan_all(const std::vector<uint8_t>& b) {
  const int64_t len = (int64_t)b.size();
  // exact-size copy: reads past `len` land in an ASan red zone
  uint8_t* buf = (uint8_t*)std::malloc(b.size() ? b.size() : 1);
  if (b.size()) std::memcpy(buf, b.data(), b.size());
  std::vector<ddrl_eb_array> arr(4);
  int32_t n = 0;
  int32_t s = ddrl_eb_scan(buf, len, arr.data(), 4, &n);
  if (s == DDRL_OK) {
    for (int k = 0; k < n && k < 4; ++k) {
      REQUIRE(arr[k].data_offset >= 0 && arr[k].nbytes >= 0 && arr[k].data_offset + arr[k].nbytes <= len);
      volatile uint8_t sink = 0;
      for (int64_t q = 0; q < arr[k].nbytes; ++q) sink ^= buf[arr[k].data_offset + q];  // the described payload is readable
      (void)sink;
    }
  }
  std::vector<ddrl_eb_msg> msgs(3);
  s = ddrl_eb_scan_forward_states(buf, len, msgs.data(), 3, &n);
  if (s == DDRL_OK)
    for (int k = 0; k < n && k < 3; ++k) REQUIRE(msgs[k].payload_offset + msgs[k].payload_len <= len);
  int64_t so, sl, oo, ol, to;
  s = ddrl_eb_scan_backward(buf, len, &so, &sl, &oo, &ol, &to);
  if (s == DDRL_OK) REQUIRE(so + sl <= len && oo + ol <= len && to <= len && to >= 16);
  const int64_t cap = (int64_t)(rnd() % 4096);
  uint8_t* dst = (uint8_t*)std::malloc(cap ? cap : 1);
  int64_t ns = 0, per = 0;
  s = ddrl_eb_frames_to_u8(buf, len, (int32_t)(rnd() % 3), dst, cap, &ns, &per);
  if (s == DDRL_OK) REQUIRE(ns >= 0 && per >= 0 && (ns == 0 || per <= cap));
  std::free(dst);
  std::free(buf);
}

int main(int argc, char** argv) {
  state = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1;
  const long iters = argc > 2 ? std::strtol(argv[2], nullptr, 10) : 20000;
  long roundtrips = 0, rejected = 0;
  for (long it = 0; it < iters; ++it) {
    // ---- (1) a valid forward-states message: header + 1..3 arrays ------------------------------------
    std::vector<uint8_t> payload;
    const int n_arr = 1 + (int)(rnd() % 3);
    const int64_t lead = 1 + (int64_t)(rnd() % 3);  // shared leading dim (samples)
    for (int a = 0; a < n_arr; ++a) {
      const int32_t dtype = 1 + (int32_t)(rnd() % 4);
      const int32_t ndim = 1 + (int32_t)(rnd() % 4);
      int64_t dims[8], count = 1;
      for (int d = 0; d < ndim; ++d) { dims[d] = d == 0 ? lead : (int64_t)(rnd() % 6); count *= dims[d]; }
      const int isz = dtype == 1 ? 1 : dtype == 2 ? 2 : dtype == 3 ? 4 : 8;
      std::vector<uint8_t> data((size_t)(count * isz) + 1);
      for (auto& x : data) x = (uint8_t)rnd();
      if (dtype == 4)  // float64 frames hold u8 / 255.0 on the wire: keep them finite so that the u8 mapping is meaningful
        for (int64_t k = 0; k < count; ++k) { double v = (double)(rnd() % 256) / 255.0; std::memcpy(&data[8 * k], &v, 8); }
      int64_t need = 0, wrote = 0;
      REQUIRE(ddrl_eb_array_bytes(dtype, ndim, dims, &need) == DDRL_OK);
      const size_t at = payload.size();
      payload.resize(at + (size_t)need);
      REQUIRE(ddrl_eb_encode_array(dtype, ndim, dims, data.data(), payload.data() + at, need, &wrote) == DDRL_OK && wrote == need);
      REQUIRE(ddrl_eb_encode_array(dtype, ndim, dims, data.data(), payload.data() + at, need - 1, &wrote) == DDRL_ERR_WORKSPACE);
      ddrl_eb_array one;
      int32_t n = 0;
      REQUIRE(ddrl_eb_scan(payload.data() + at, need, &one, 1, &n) == DDRL_OK && n == 1);
      REQUIRE(one.dtype == dtype && one.ndim == ndim && one.count == count && one.nbytes == count * isz);
      REQUIRE(std::memcmp(payload.data() + at + one.data_offset, data.data(), (size_t)one.nbytes) == 0);
      ++roundtrips;
    }
    std::vector<uint8_t> msg(20);
    const int32_t ip[4] = {(int32_t)(rnd() % 256), 0, 1, 2};
    REQUIRE(ddrl_eb_forward_header(ip, (uint32_t)rnd(), (uint64_t)payload.size(), msg.data()) == DDRL_OK);
    msg.insert(msg.end(), payload.begin(), payload.end());
    scan_all(msg);
    scan_all(payload);
    // a backward blob around the same arrays: >Q len + states, >Q len + others, tail
    std::vector<uint8_t> blob;
    put64(blob, payload.size());
    blob.insert(blob.end(), payload.begin(), payload.end());
    put64(blob, payload.size());
    blob.insert(blob.end(), payload.begin(), payload.end());
    for (int t = (int)(rnd() % 9); t > 0; --t) blob.push_back((uint8_t)rnd());
    scan_all(blob);
    // ---- (2) truncations and byte mutations -----------------------------------------------------------
    for (std::vector<uint8_t>* src : {&msg, &payload, &blob}) {
      std::vector<uint8_t> cut(src->begin(), src->begin() + (long)(rnd() % (src->size() + 1)));
      scan_all(cut);
      std::vector<uint8_t> mut(*src);
      for (int m = 1 + (int)(rnd() % 4); m > 0 && !mut.empty(); --m) {
        const size_t at = (size_t)(rnd() % mut.size());
        switch (rnd() % 4) {
          case 0: mut[at] = (uint8_t)rnd(); break;
          case 1: mut[at] ^= (uint8_t)(1u << (rnd() % 8)); break;
          case 2: mut[at] = 0xFF; break;
          default: mut[at] = 0; break;
        }
      }
      scan_all(mut);
      ++rejected;
    }
    // ---- (3) hostile headers -----------------------------------------------------------------------------
    {
      std::vector<uint8_t> rec = {0, (uint8_t)(1 + rnd() % 4)};
      put32(rec, (uint32_t)rnd());
      const uint32_t ndim = 2 + (uint32_t)(rnd() % 7);
      put32(rec, ndim);
      for (uint32_t d = 0; d < ndim; ++d) put32(rec, rnd() % 3 ? (uint32_t)rnd() | 0x80000000u : (uint32_t)(rnd() % 5));
      for (int t = (int)(rnd() % 64); t > 0; --t) rec.push_back((uint8_t)rnd());
      scan_all(rec);
      int64_t dims[8], nb = 0;
      for (auto& d : dims) d = (int64_t)(rnd() >> (rnd() % 40));
      (void)ddrl_eb_array_bytes(1 + (int32_t)(rnd() % 4), (int32_t)(rnd() % 10) - 1, dims, &nb);
      std::vector<uint8_t> hdr;
      const uint64_t big[5] = {0x7FFFFFFFFFFFFFF0ull, 0x7FFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFF0ull, 0x8000000000000000ull, rnd()};
      put64(hdr, big[rnd() % 5]);
      put64(hdr, big[rnd() % 5]);
      put32(hdr, (uint32_t)rnd());
      for (int t = (int)(rnd() % 32); t > 0; --t) hdr.push_back((uint8_t)rnd());
      scan_all(hdr);
    }
  }
  // csrc/comm.cpp without a communicator: the argument checks come before anything touches RCCL
  uint8_t id[128] = {0};
  ddrl_comm* c = nullptr;
  REQUIRE(ddrl_comm_unique_id(nullptr) == DDRL_ERR_INVALID_ARG);
  REQUIRE(ddrl_comm_info(nullptr, 8, nullptr) == DDRL_ERR_INVALID_ARG);
  REQUIRE(ddrl_comm_info((char*)id, -1, nullptr) == DDRL_ERR_INVALID_ARG);
  REQUIRE(ddrl_comm_create(nullptr, 0, 1, &c) == DDRL_ERR_INVALID_ARG);
  REQUIRE(ddrl_comm_create(id, 2, 2, &c) == DDRL_ERR_INVALID_ARG);
  REQUIRE(ddrl_comm_create(id, 0, 0, &c) == DDRL_ERR_INVALID_ARG);
  REQUIRE(ddrl_comm_destroy(nullptr) == DDRL_ERR_INVALID_ARG);
  float f = 0.0f;
  REQUIRE(ddrl_allreduce_f32(nullptr, &f, 1, nullptr) == DDRL_ERR_INVALID_ARG);
  REQUIRE(ddrl_broadcast_f32(nullptr, &f, 1, 0, nullptr) == DDRL_ERR_INVALID_ARG);
  std::printf("eb_fuzz OK: %ld iterations, %ld round trips, %ld mutated streams\n", iters, roundtrips, rejected * 2);
  return 0;
}
