#include "weights.h"

#include <cstring>
#include <stdexcept>

namespace pe {

static const char MAGIC[8] = {'P', 'E', 'B', 'L', 'O', 'B', '0', '1'};
static constexpr size_t NAME_BYTES = 96;
static constexpr size_t REC_BYTES = NAME_BYTES + 4 + 16 + 4 + 8 + 8;

const HostTensor& WeightSet::get(const std::string& name) const {
  auto it = t.find(name);
  if (it == t.end()) throw std::runtime_error("voice is missing tensor '" + name + "'");
  return it->second;
}

void WeightSet::put(const std::string& name, HostTensor&& ht) {
  if (!t.count(name)) order.push_back(name);
  t[name] = std::move(ht);
}

size_t blob_header_bytes(const void* data, size_t nbytes) {
  const uint8_t* p = static_cast<const uint8_t*>(data);
  const size_t head = 8 + 4 * ARCH_INTS + 8;
  if (nbytes < head || memcmp(p, MAGIC, 8) != 0) throw std::runtime_error("not a PEBLOB01 weight blob");
  uint32_t n;
  memcpy(&n, p + 8 + 4 * ARCH_INTS, 4);
  if (nbytes < head + (size_t)n * REC_BYTES) throw std::runtime_error("truncated weight blob (records)");
  return head + (size_t)n * REC_BYTES;
}

WeightSet parse_blob(const void* data, size_t nbytes, bool shapes_only) {
  const uint8_t* p = static_cast<const uint8_t*>(data);
  const size_t head = 8 + 4 * ARCH_INTS + 8;
  if (nbytes < head || memcmp(p, MAGIC, 8) != 0) throw std::runtime_error("not a PEBLOB01 weight blob");
  WeightSet ws;
  memcpy(ws.arch, p + 8, 4 * ARCH_INTS);
  uint32_t n;
  memcpy(&n, p + 8 + 4 * ARCH_INTS, 4);
  if (nbytes < head + (size_t)n * REC_BYTES) throw std::runtime_error("truncated weight blob (records)");
  const uint8_t* r = p + head;
  for (uint32_t i = 0; i < n; ++i, r += REC_BYTES) {
    char name[NAME_BYTES + 1];
    memcpy(name, r, NAME_BYTES);
    name[NAME_BYTES] = 0;
    int32_t ndim, dims[4];
    uint64_t off, numel;
    memcpy(&ndim, r + NAME_BYTES, 4);
    memcpy(dims, r + NAME_BYTES + 4, 16);
    memcpy(&off, r + NAME_BYTES + 24, 8);
    memcpy(&numel, r + NAME_BYTES + 32, 8);
    if (ndim < 0 || ndim > 4) throw std::runtime_error("bad tensor rank in blob");
    // overflow-safe: a corrupt blob (also one received over the weight broadcast) must not wrap the bound
    if (!shapes_only && (off > nbytes || numel > (nbytes - off) / 4)) throw std::runtime_error("truncated weight blob (data)");
    if (numel > ((uint64_t)1 << 34)) throw std::runtime_error("implausible tensor size in blob");
    HostTensor ht;
    uint64_t chk = 1;
    for (int d = 0; d < ndim; ++d) {
      ht.dims.push_back(dims[d]);
      chk *= (uint64_t)dims[d];
    }
    if (chk != numel) throw std::runtime_error(std::string("dims/numel mismatch for ") + name);
    if (!shapes_only) {
      ht.data.resize(numel);
      memcpy(ht.data.data(), p + off, numel * 4);
    }
    ws.put(name, std::move(ht));
  }
  return ws;
}

std::vector<uint8_t> serialize_blob(const WeightSet& ws) {
  const size_t n = ws.order.size();
  size_t head = 8 + 4 * ARCH_INTS + 8 + n * REC_BYTES;
  size_t off = (head + 63) / 64 * 64;
  std::vector<size_t> offs(n);
  for (size_t i = 0; i < n; ++i) {
    offs[i] = off;
    off = (off + ws.t.at(ws.order[i]).data.size() * 4 + 63) / 64 * 64;
  }
  std::vector<uint8_t> out(off, 0);
  memcpy(out.data(), MAGIC, 8);
  memcpy(out.data() + 8, ws.arch, 4 * ARCH_INTS);
  uint32_t n32 = (uint32_t)n;
  memcpy(out.data() + 8 + 4 * ARCH_INTS, &n32, 4);
  uint8_t* r = out.data() + 8 + 4 * ARCH_INTS + 8;
  for (size_t i = 0; i < n; ++i, r += REC_BYTES) {
    const std::string& name = ws.order[i];
    const HostTensor& ht = ws.t.at(name);
    if (name.size() >= NAME_BYTES) throw std::runtime_error("tensor name too long: " + name);
    memcpy(r, name.data(), name.size());
    int32_t ndim = (int32_t)ht.dims.size(), dims[4] = {1, 1, 1, 1};
    for (int d = 0; d < ndim; ++d) dims[d] = (int32_t)ht.dims[d];
    uint64_t o64 = offs[i], numel = ht.data.size();
    memcpy(r + NAME_BYTES, &ndim, 4);
    memcpy(r + NAME_BYTES + 4, dims, 16);
    memcpy(r + NAME_BYTES + 24, &o64, 8);
    memcpy(r + NAME_BYTES + 32, &numel, 8);
    memcpy(out.data() + offs[i], ht.data.data(), numel * 4);
  }
  return out;
}

}  // namespace pe
