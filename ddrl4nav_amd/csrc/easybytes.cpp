// EasyBytes wire codec (host side) -- SURVEY.md section 8f row 1.
//
// Byte-exact re-implementation of the reference's tagged-array codec and message framings
// (USTC_lab/data/easybytes.py:18-172) so that existing env workers / servers can talk to the
// HIP path over the unchanged Redis protocol:
//   array record   : >h dtype code (1 u8, 2 f16, 3 f32, 4 f64)   easybytes.py:21-26,33-45
//                    >I count, >I ndim, ndim x >I dims, raw little-endian payload   :63-75
//   forward states : >Q payload length, 4 x >H ip parts, >I process_env_id, arrays   :141-149
//   backward blob  : >Q len(states) + states arrays, >Q len(other4) + arrays, marshal tail :151-162
// plus the one hot helper of this row: turning the float64/float32 frames the reference ships
// (uint8/255.0, warputils.py:300) back into the exact uint8 bytes, straight into a pinned ring
// slot (data/ring.py) that hipMemcpyAsync then moves to the device pool.
// Pure host code: nothing here touches the GPU.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "../../include/ddrl.h"

namespace {

inline uint16_t be16(const uint8_t* p) { return (uint16_t)((p[0] << 8) | p[1]); }
inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline uint64_t be64(const uint8_t* p) { return ((uint64_t)be32(p) << 32) | be32(p + 4); }
inline void put16(uint8_t* p, uint16_t v) { p[0] = (uint8_t)(v >> 8); p[1] = (uint8_t)v; }
inline void put32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }
inline void put64(uint8_t* p, uint64_t v) { put32(p, (uint32_t)(v >> 32)); put32(p + 4, (uint32_t)v); }

inline int item_size(int32_t dtype) {
  switch (dtype) {
    case 1: return 1;
    case 2: return 2;
    case 3: return 4;
    case 4: return 8;
    default: return 0;
  }
}

inline float half_to_float(uint16_t h) {
  const uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31u, m = h & 1023u;
  uint32_t bits;
  if (e == 0) {
    if (m == 0) bits = s;
    else {
      int sh = 0;
      uint32_t mm = m;
      while (!(mm & 1024u)) { mm <<= 1; ++sh; }
      bits = s | ((uint32_t)(113 - sh) << 23) | ((mm & 1023u) << 13);
    }
  } else if (e == 31) bits = s | 0x7F800000u | (m << 13);
  else bits = s | ((e + 112u) << 23) | (m << 13);
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}

}  // namespace

extern "C" {

int32_t ddrl_eb_array_bytes(int32_t dtype, int32_t ndim, const int64_t* dims, int64_t* nbytes) {
  const int isz = item_size(dtype);
  if (!isz || ndim < 0 || ndim > 8 || (ndim && !dims) || !nbytes) return DDRL_ERR_INVALID_ARG;
  int64_t count = 1;
  for (int i = 0; i < ndim; ++i) {
    if (dims[i] < 0 || dims[i] > 0xFFFFFFFFll) return DDRL_ERR_INVALID_ARG;
    // the wire format stores count as uint32: test BEFORE multiplying (two 32-bit factors can pass INT64_MAX)
    if (dims[i] != 0 && count > 0xFFFFFFFFll / dims[i]) return DDRL_ERR_INVALID_ARG;
    count *= dims[i];
  }
  *nbytes = 2 + 8 + 4 * (int64_t)ndim + count * isz;  // count <= 2^32 - 1, isz <= 8: no wrap
  return DDRL_OK;
}

int32_t ddrl_eb_encode_array(int32_t dtype, int32_t ndim, const int64_t* dims, const void* data, uint8_t* out, int64_t cap,
                             int64_t* written) {
  int64_t need = 0;
  int32_t s = ddrl_eb_array_bytes(dtype, ndim, dims, &need);
  if (s != DDRL_OK) return s;
  if (!out || !written || (!data && need > 10 + 4 * ndim)) return DDRL_ERR_INVALID_ARG;
  if (need > cap) return DDRL_ERR_WORKSPACE;
  int64_t count = 1;
  for (int i = 0; i < ndim; ++i) count *= dims[i];  // bounded by ddrl_eb_array_bytes above
  put16(out, (uint16_t)dtype);
  put32(out + 2, (uint32_t)count);
  put32(out + 6, (uint32_t)ndim);
  for (int i = 0; i < ndim; ++i) put32(out + 10 + 4 * i, (uint32_t)dims[i]);
  std::memcpy(out + 10 + 4 * ndim, data, (size_t)(count * item_size(dtype)));
  *written = need;
  return DDRL_OK;
}

// decode_data without copying: walk the records of buf[0:len) and describe each one.
int32_t ddrl_eb_scan(const uint8_t* buf, int64_t len, ddrl_eb_array* out, int32_t cap, int32_t* n) {
  if (!buf || len < 0 || !n || (cap > 0 && !out)) return DDRL_ERR_INVALID_ARG;
  int64_t i = 0;
  int32_t k = 0;
  while (i < len) {
    if (len - i < 10) return DDRL_ERR_INVALID_ARG;
    const int32_t dtype = (int16_t)be16(buf + i);
    const int isz = item_size(dtype);
    if (!isz) return DDRL_ERR_UNSUPPORTED;  // "Match data type error" -> ValueError in the reference
    const int64_t count = be32(buf + i + 2);
    const int32_t ndim = (int32_t)be32(buf + i + 6);
    if (ndim < 0 || ndim > 8 || 10 + 4 * (int64_t)ndim > len - i) return DDRL_ERR_INVALID_ARG;
    int64_t prod = 1;
    ddrl_eb_array a;
    std::memset(&a, 0, sizeof(a));
    for (int d = 0; d < ndim; ++d) {
      a.dims[d] = be32(buf + i + 10 + 4 * d);
      // count is a uint32 on the wire.  The test comes BEFORE the product: prod and the dim are both < 2^32, so prod * dim can reach
      // 1.8e19 > INT64_MAX (signed overflow, undefined) if multiplied first
      if (a.dims[d] != 0 && prod > 0xFFFFFFFFll / a.dims[d]) return DDRL_ERR_INVALID_ARG;
      prod *= a.dims[d];
    }
    if (prod != count) return DDRL_ERR_INVALID_ARG;
    a.dtype = dtype;
    a.ndim = ndim;
    a.count = count;
    a.data_offset = i + 10 + 4 * ndim;
    a.nbytes = count * isz;
    if (a.nbytes > len - a.data_offset) return DDRL_ERR_INVALID_ARG;
    if (k < cap) out[k] = a;
    ++k;
    i = a.data_offset + a.nbytes;
  }
  *n = k;
  return k > cap && cap > 0 ? DDRL_ERR_WORKSPACE : DDRL_OK;
}

// 20-byte header of encode_forward_states: >Q length, 4 x >H ip, >I process_env_id
int32_t ddrl_eb_forward_header(const int32_t ip[4], uint32_t process_env_id, uint64_t payload_len, uint8_t* out20) {
  if (!ip || !out20) return DDRL_ERR_INVALID_ARG;
  for (int i = 0; i < 4; ++i)
    if (ip[i] < 0 || ip[i] > 255) return DDRL_ERR_INVALID_ARG;
  put64(out20, payload_len);
  for (int i = 0; i < 4; ++i) put16(out20 + 8 + 2 * i, (uint16_t)ip[i]);
  put32(out20 + 16, process_env_id);
  return DDRL_OK;
}

// Walk the concatenated forward-states messages of one BLPOP item (decode_forward_states).
int32_t ddrl_eb_scan_forward_states(const uint8_t* buf, int64_t len, ddrl_eb_msg* out, int32_t cap, int32_t* n) {
  if (!buf || len < 0 || !n || (cap > 0 && !out)) return DDRL_ERR_INVALID_ARG;
  int64_t i = 0;
  int32_t k = 0;
  while (i < len) {
    if (len - i < 20) return DDRL_ERR_INVALID_ARG;
    ddrl_eb_msg m;
    m.payload_len = (int64_t)be64(buf + i);
    for (int q = 0; q < 4; ++q) m.ip[q] = be16(buf + i + 8 + 2 * q);
    m.process_env_id = be32(buf + i + 16);
    m.payload_offset = i + 20;
    // the length is an untrusted u64: compare without adding, so nothing wraps
    if (m.payload_len < 0 || m.payload_len > len - m.payload_offset) return DDRL_ERR_INVALID_ARG;
    if (k < cap) out[k] = m;
    ++k;
    i = m.payload_offset + m.payload_len;
  }
  *n = k;
  return k > cap && cap > 0 ? DDRL_ERR_WORKSPACE : DDRL_OK;
}

// Frames of a (batched) forward-states item -> contiguous uint8 in dst (e.g. a pinned ring slot):
// array `state_index` of every message, concatenated in message order (np.concatenate axis 0 of
// easybytes.py:132-137).  u8 payloads are copied; f64/f32/f16 payloads hold uint8/255.0 and are
// mapped back with round(x*255) (exact: |x*255 - k| < 2^-20).
int32_t ddrl_eb_frames_to_u8(const uint8_t* buf, int64_t len, int32_t state_index, uint8_t* dst, int64_t dst_cap,
                             int64_t* n_samples, int64_t* sample_elems) {
  if (!buf || !dst || !n_samples || !sample_elems || state_index < 0 || len < 0 || dst_cap < 0) return DDRL_ERR_INVALID_ARG;
  int64_t i = 0, written = 0, total = 0, per = -1;
  while (i < len) {
    if (len - i < 20) return DDRL_ERR_INVALID_ARG;
    const int64_t plen = (int64_t)be64(buf + i);
    const int64_t poff = i + 20;
    if (plen < 0 || plen > len - poff) return DDRL_ERR_INVALID_ARG;
    ddrl_eb_array arr[16];
    int32_t na = 0;
    int32_t s = ddrl_eb_scan(buf + poff, plen, arr, 16, &na);
    if (s != DDRL_OK) return s;
    if (state_index >= na) return DDRL_ERR_INVALID_ARG;
    const ddrl_eb_array& a = arr[state_index];
    if (a.ndim < 1) return DDRL_ERR_INVALID_ARG;
    const int64_t elems = a.dims[0] ? a.count / a.dims[0] : 0;
    if (per < 0) per = elems;
    if (elems != per) return DDRL_ERR_INVALID_ARG;
    if (a.count > dst_cap - written) return DDRL_ERR_WORKSPACE;
    const uint8_t* src = buf + poff + a.data_offset;
    uint8_t* d = dst + written;
    switch (a.dtype) {
      case 1: std::memcpy(d, src, (size_t)a.count); break;
      case 4:
        for (int64_t k = 0; k < a.count; ++k) {
          double v;
          std::memcpy(&v, src + 8 * k, 8);
          d[k] = (uint8_t)std::lrint(v * 255.0);
        }
        break;
      case 3:
        for (int64_t k = 0; k < a.count; ++k) {
          float v;
          std::memcpy(&v, src + 4 * k, 4);
          d[k] = (uint8_t)std::lrintf(v * 255.0f);
        }
        break;
      case 2:
        for (int64_t k = 0; k < a.count; ++k) {
          uint16_t h;
          std::memcpy(&h, src + 2 * k, 2);
          d[k] = (uint8_t)std::lrintf(half_to_float(h) * 255.0f);
        }
        break;
      default: return DDRL_ERR_UNSUPPORTED;
    }
    written += a.count;
    total += a.dims[0];
    i = poff + plen;
  }
  *n_samples = total;
  *sample_elems = per < 0 ? 0 : per;
  return DDRL_OK;
}

// Sections of a backward blob (decode_backward_data): states arrays, other-4 arrays, marshal tail.
int32_t ddrl_eb_scan_backward(const uint8_t* buf, int64_t len, int64_t* states_off, int64_t* states_len, int64_t* other_off,
                              int64_t* other_len, int64_t* tail_off) {
  if (!buf || !states_off || !states_len || !other_off || !other_len || !tail_off || len < 16) return DDRL_ERR_INVALID_ARG;
  const int64_t sl = (int64_t)be64(buf);
  if (sl < 0 || sl > len - 16) return DDRL_ERR_INVALID_ARG;
  const int64_t ol = (int64_t)be64(buf + 8 + sl);
  if (ol < 0 || ol > len - 16 - sl) return DDRL_ERR_INVALID_ARG;
  *states_off = 8;
  *states_len = sl;
  *other_off = 16 + sl;
  *other_len = ol;
  *tail_off = 16 + sl + ol;
  return DDRL_OK;
}

int32_t ddrl_eb_put_u64(uint64_t v, uint8_t* out8) {
  if (!out8) return DDRL_ERR_INVALID_ARG;
  put64(out8, v);
  return DDRL_OK;
}

}  // extern "C"
