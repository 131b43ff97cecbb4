#include "weights.h"

#include <cstdarg>
#include <cstring>
#include <fstream>

namespace sd {

thread_local std::string g_last_error;

void fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  throw Error(code, buf);
}

// ---- Arena: chunked bump allocator (chunks of >= 256 MiB; HBM is 288 GB, nothing is recycled) ----
Arena::~Arena() {
  for (void* p : chunks_) (void)hipFree(p);
}

void* Arena::alloc(size_t bytes) {
  bytes = align_up(bytes ? bytes : 1, 256);
  if (chunks_.empty() || cur_ + bytes > cap_) {
    size_t chunk = std::max<size_t>(bytes, (size_t)256 << 20);
    void* p = nullptr;
    SD_HIP(hipMalloc(&p, chunk));
    SD_HIP(hipMemset(p, 0, chunk));
    SD_HIP(hipDeviceSynchronize());   // the memset runs on the null stream; handles use their own
    chunks_.push_back(p);
    sizes_.push_back(chunk);
    cap_ = chunk;
    cur_ = 0;
    total_ += chunk;
  }
  char* p = reinterpret_cast<char*>(chunks_.back()) + cur_;
  cur_ += bytes;
  return p;
}

// ---- WeightStore ------------------------------------------------------------------------------
static inline float half_bits_to_float(uint16_t h) {
  _Float16 v;
  std::memcpy(&v, &h, 2);
  return (float)v;
}
static inline float bf16_bits_to_float(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

void WeightStore::add(const std::string& name, const void* data, int dtype, const int64_t* shape, int ndim) {
  SD_REQUIRE(data && ndim >= 0 && ndim <= 8, kInvalidArgument, "weights.add(%s): bad arguments", name.c_str());
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  size_t n = t.numel();
  t.data.resize(n);
  if (dtype == 1) {
    std::memcpy(t.data.data(), data, n * 4);
  } else if (dtype == 0) {
    const uint16_t* s = reinterpret_cast<const uint16_t*>(data);
    for (size_t i = 0; i < n; ++i) t.data[i] = half_bits_to_float(s[i]);
  } else if (dtype == 2) {
    const uint16_t* s = reinterpret_cast<const uint16_t*>(data);
    for (size_t i = 0; i < n; ++i) t.data[i] = bf16_bits_to_float(s[i]);
  } else {
    fail(kInvalidArgument, "weights.add(%s): unknown dtype %d", name.c_str(), dtype);
  }
  map_[name] = std::move(t);
}

// Public SD 1.x / 2.x VAE checkpoints keep the deprecated attention names (query / key / value /
// proj_attn) that diffusers renames at load time; look those up when the current name is absent.
const HostTensor* WeightStore::find(const std::string& name) const {
  auto it = map_.find(name);
  if (it != map_.end()) return &it->second;
  static const char* const kAlias[][2] = {
      {".to_q.", ".query."}, {".to_k.", ".key."}, {".to_v.", ".value."}, {".to_out.0.", ".proj_attn."}};
  for (const auto& a : kAlias) {
    const size_t pos = name.find(a[0]);
    if (pos == std::string::npos) continue;
    std::string old = name;
    old.replace(pos, std::strlen(a[0]), a[1]);
    it = map_.find(old);
    if (it != map_.end()) return &it->second;
  }
  return nullptr;
}

const HostTensor& WeightStore::get(const std::string& name) const {
  const HostTensor* t = find(name);
  if (!t) fail(kNotFound, "checkpoint is missing tensor '%s'", name.c_str());
  return *t;
}

// Minimal safetensors reader: 8-byte LE header length, JSON header
//   {"name": {"dtype": "F16", "shape": [..], "data_offsets": [b, e]}, ..., "__metadata__": {...}}
// followed by the raw little-endian tensor bytes.  The header grammar is a flat two-level object,
// so a small hand-rolled scanner is enough (no JSON dependency).
namespace {
struct Scanner {
  const std::string& s;
  size_t i = 0;
  explicit Scanner(const std::string& str) : s(str) {}
  void ws() {
    while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) ++i;
  }
  bool eat(char c) {
    ws();
    if (i < s.size() && s[i] == c) {
      ++i;
      return true;
    }
    return false;
  }
  void expect(char c) {
    if (!eat(c)) fail(kInvalidArgument, "safetensors header: expected '%c' at byte %zu", c, i);
  }
  std::string str() {
    expect('"');
    std::string out;
    while (i < s.size() && s[i] != '"') {
      if (s[i] == '\\' && i + 1 < s.size()) ++i;
      out.push_back(s[i++]);
    }
    expect('"');
    return out;
  }
  int64_t num() {
    ws();
    size_t j = i;
    while (j < s.size() && (isdigit((unsigned char)s[j]) || s[j] == '-')) ++j;
    if (j == i || j - i > 18) fail(kInvalidArgument, "safetensors header: expected a number at byte %zu", i);
    int64_t v = 0;
    try {
      v = std::stoll(s.substr(i, j - i));
    } catch (const std::exception&) {
      fail(kInvalidArgument, "safetensors header: bad number at byte %zu", i);
    }
    i = j;
    return v;
  }
  void skip_value() {   // for __metadata__
    ws();
    if (i >= s.size()) fail(kInvalidArgument, "safetensors header: truncated");
    if (s[i] == '{') {
      int depth = 0;
      bool in_str = false;
      for (; i < s.size(); ++i) {
        if (in_str) {
          if (s[i] == '\\') ++i;
          else if (s[i] == '"') in_str = false;
        } else if (s[i] == '"') in_str = true;
        else if (s[i] == '{') ++depth;
        else if (s[i] == '}' && --depth == 0) {
          ++i;
          return;
        }
      }
    } else if (s[i] == '"') {
      str();
    } else {
      while (i < s.size() && s[i] != ',' && s[i] != '}') ++i;
    }
  }
};
}  // namespace

void WeightStore::load_safetensors(const std::string& path, const std::string& prefix) {
  std::ifstream f(path, std::ios::binary);
  if (!f) fail(kNotFound, "%s not found", path.c_str());
  uint64_t hlen = 0;
  f.read(reinterpret_cast<char*>(&hlen), 8);
  SD_REQUIRE(f && hlen > 0 && hlen < ((uint64_t)1 << 30), kInvalidArgument, "%s: bad safetensors header", path.c_str());
  f.seekg(0, std::ios::end);
  const uint64_t file_size = (uint64_t)f.tellg();
  SD_REQUIRE(8 + hlen <= file_size, kInvalidArgument, "%s: safetensors header (%llu bytes) exceeds the file", path.c_str(),
             (unsigned long long)hlen);
  f.seekg(8);
  std::string header(hlen, '\0');
  f.read(&header[0], (std::streamsize)hlen);
  SD_REQUIRE((bool)f && (uint64_t)f.gcount() == hlen, kInvalidArgument, "%s: truncated safetensors header", path.c_str());
  const uint64_t data_start = 8 + hlen;
  Scanner sc(header);
  sc.expect('{');
  std::vector<char> raw;
  while (true) {
    if (sc.eat('}')) break;
    std::string key = sc.str();
    sc.expect(':');
    if (key == "__metadata__") {
      sc.skip_value();
      sc.eat(',');
      continue;
    }
    std::string dtype;
    std::vector<int64_t> shape;
    int64_t b = 0, e = 0;
    sc.expect('{');
    while (true) {
      std::string field = sc.str();
      sc.expect(':');
      if (field == "dtype") {
        dtype = sc.str();
      } else if (field == "shape") {
        sc.expect('[');
        while (!sc.eat(']')) {
          shape.push_back(sc.num());
          sc.eat(',');
        }
      } else if (field == "data_offsets") {
        sc.expect('[');
        b = sc.num();
        sc.expect(',');
        e = sc.num();
        sc.expect(']');
      } else {
        sc.skip_value();
      }
      if (sc.eat('}')) break;
      sc.expect(',');
    }
    sc.eat(',');
    int dt = dtype == "F16" ? 0 : dtype == "F32" ? 1 : dtype == "BF16" ? 2 : -1;
    if (dt < 0) continue;   // integer buffers etc. are not part of the path
    // never trust the header: offsets inside the file, non-negative dims, byte count == numel * element size
    SD_REQUIRE(b >= 0 && e >= b && data_start + (uint64_t)e <= file_size, kInvalidArgument,
               "%s: tensor '%s' has data_offsets [%lld, %lld) outside the file", path.c_str(), key.c_str(), (long long)b,
               (long long)e);
    SD_REQUIRE(shape.size() <= 8, kInvalidArgument, "%s: tensor '%s' has %zu dims", path.c_str(), key.c_str(), shape.size());
    uint64_t numel = 1;
    for (int64_t dim : shape) {
      SD_REQUIRE(dim >= 0 && (dim == 0 || numel <= ((uint64_t)1 << 40) / (uint64_t)dim), kInvalidArgument,
                 "%s: tensor '%s' has a bad dimension %lld", path.c_str(), key.c_str(), (long long)dim);
      numel *= (uint64_t)dim;
    }
    SD_REQUIRE((uint64_t)(e - b) == numel * (dt == 1 ? 4u : 2u), kInvalidArgument,
               "%s: tensor '%s' stores %lld bytes but its shape needs %llu", path.c_str(), key.c_str(), (long long)(e - b),
               (unsigned long long)(numel * (dt == 1 ? 4u : 2u)));
    raw.resize((size_t)(e - b));
    f.seekg((std::streamoff)(data_start + (uint64_t)b));
    f.read(raw.data(), (std::streamsize)raw.size());
    SD_REQUIRE((bool)f, kInvalidArgument, "%s: truncated tensor '%s'", path.c_str(), key.c_str());
    std::string name = key;
    if (!prefix.empty() && name.compare(0, prefix.size(), prefix) == 0) name = name.substr(prefix.size());
    add(name, raw.data(), dt, shape.data(), (int)shape.size());
  }
}

}  // namespace sd
