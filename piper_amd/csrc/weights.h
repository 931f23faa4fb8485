// Host-side voice weights: canonical tensor set + architecture ints. Mirrors piper_amd/weights.py
// (same arch[] indices, same PEBLOB01 layout).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace pe {

enum {
  A_NVOCAB = 0, A_HIDDEN, A_INTER, A_FILTER, A_NHEADS, A_NLAYERS, A_KSIZE, A_WINDOW, A_RESBLOCK, A_NRB,
  A_RBK0 = 10, A_NDIL = 14, A_RBDIL0 = 15, A_NUPS = 31, A_UPR0 = 32, A_UPK0 = 40, A_UPINIT = 48,
  A_NSPK = 49, A_GIN = 50, A_SR = 51, A_DPFLOWS = 52, A_DDSLAYERS = 53, A_NBINS = 54, A_FLOWN = 55,
  A_WNLAYERS = 56, A_WNK = 57, ARCH_INTS = 64
};
static constexpr int MAX_DIL = 4;

struct HostTensor {
  std::vector<int64_t> dims;
  std::vector<float> data;
  int64_t numel() const {
    int64_t n = 1;
    for (auto d : dims) n *= d;
    return n;
  }
};

struct WeightSet {
  int32_t arch[ARCH_INTS] = {0};
  std::map<std::string, HostTensor> t;
  std::vector<std::string> order;      // insertion order (blob order)

  const HostTensor& get(const std::string& name) const;
  bool has(const std::string& name) const { return t.count(name) != 0; }
  void put(const std::string& name, HostTensor&& ht);
};

// shapes_only: read the architecture header and the tensor records (names, dims) but no data -- `data` may then be
// just the header part of a blob (what the ranks of a multi-GPU job receive before the packed weights arrive).
WeightSet parse_blob(const void* data, size_t nbytes, bool shapes_only = false);
size_t blob_header_bytes(const void* data, size_t nbytes);
std::vector<uint8_t> serialize_blob(const WeightSet& ws);

// onnx_reader.cpp: reads a Piper voice .onnx (export_onnx.py graph) and recovers the canonical tensors.
WeightSet load_onnx(const std::string& path);

}  // namespace pe
