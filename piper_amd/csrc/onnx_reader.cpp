#include "weights.h"
#include <stdexcept>
namespace pe {
WeightSet load_onnx(const std::string& path) { throw std::runtime_error("onnx loader not built yet: " + path); }
}
