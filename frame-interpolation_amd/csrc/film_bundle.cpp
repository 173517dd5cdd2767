// film_bundle.cpp -- film_load_bundle: the variable-restore half of `tf.compat.v2.saved_model.load(model_path)`
// (reference eval/interpolator.py:148; the files are written by model.save(), training/train_lib.py:280 and
// training/build_saved_model_cli.py:65-73) without TensorFlow and without Python: a non-Python host hands the SavedModel
// directory to the C-ABI and gets a finalized handle.
//
// Formats, restated from their published definitions (same statement as film_hip/tf_bundle.py, the Python twin the tests keep
// as a second reader):
//   <prefix>.index   LevelDB-style sorted string table (tensorflow/core/lib/io/table_format.txt):
//                      [data block]* [metaindex block] [index block] [48-byte footer]
//                    footer = BlockHandle(metaindex) BlockHandle(index), zero padding to 40 bytes, magic 0xdb4775248b80fb57 LE;
//                    block  = prefix-compressed entries (varint32 shared, varint32 unshared, varint32 value length, key suffix,
//                             value), uint32 restart offsets, uint32 restart count; then 1 type byte (0 raw, 1 snappy) and the
//                             masked crc32c of contents + type; BlockHandle = varint64 offset, varint64 size (without trailer).
//   values           tensorflow/core/protobuf/tensor_bundle.proto: key "" -> BundleHeaderProto {num_shards = 1, endianness = 2};
//                    other keys -> BundleEntryProto {dtype = 1, shape = 2 {dim = 2 {size = 1}}, shard_id = 3, offset = 4, size = 5,
//                    crc32c = 6 (fixed32, masked), slices = 7}; DT_FLOAT = 1; tensor bytes raw row-major little-endian at
//                    [offset, offset + size) of <prefix>.data-<shard 05d>-of-<num_shards 05d>.
//   keys             Keras object-graph attribute path + "/.ATTRIBUTES/VARIABLE_VALUE"; for film_net the paths follow the attribute
//                    names of the reference source (feature_extractor.py:118-123,160; pyramid_flow_estimator.py:74-83,111-123;
//                    fusion.py:64-101) - tests/test_ref_golden_cpu.py derives them by executing that code.
// Placement rules (identical to tf_bundle.load_film_weights): 1. attribute-path patterns, independent of the layer_with_weights-N
// numbering; 2. what rule 1 left: the unused float variable of the required shape, ONLY when that shape is unique on both sides
// (film_net repeats shapes - (3,3,256,256) is three different layers - so anything ambiguous is an error, never a guess).
#include <errno.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>

#include <algorithm>
#include <set>

#include "film_internal.h"

using namespace film_internal;

namespace {

constexpr uint64_t kTableMagic = 0xdb4775248b80fb57ull;
constexpr size_t kFooterLen = 48, kBlockTrailer = 5;
const char* const kVarSuffix = "/.ATTRIBUTES/VARIABLE_VALUE";

struct Err { std::string msg; };   // thrown inside this file only, turned into fail() at the boundary

[[noreturn]] void bad(const std::string& m) { throw Err{m}; }

uint32_t unmask_crc(uint32_t masked) {
  const uint32_t rot = masked - 0xa282ead8u;
  return (rot >> 17) | (rot << 15);
}

uint64_t get_varint(const uint8_t* buf, size_t len, size_t& pos) {
  uint64_t result = 0;
  for (int shift = 0; shift <= 63; shift += 7) {
    if (pos >= len) bad("truncated varint");
    const uint8_t b = buf[pos++];
    result |= (uint64_t)(b & 0x7F) << shift;
    if (!(b & 0x80)) return result;
  }
  bad("varint too long");
}

struct PbField { uint32_t field; int wt; uint64_t v; const uint8_t* p; size_t n; };

// minimal protobuf wire decoder; length-delimited values are (p, n) views into buf
std::vector<PbField> pb_decode(const uint8_t* buf, size_t len) {
  std::vector<PbField> out;
  size_t pos = 0;
  while (pos < len) {
    const uint64_t tag = get_varint(buf, len, pos);
    PbField f{(uint32_t)(tag >> 3), (int)(tag & 7), 0, nullptr, 0};
    switch (f.wt) {
      case 0: f.v = get_varint(buf, len, pos); break;
      case 1: if (pos + 8 > len) bad("truncated fixed64"); memcpy(&f.v, buf + pos, 8); pos += 8; break;
      case 2: {
        const uint64_t n = get_varint(buf, len, pos);
        if (n > len - pos) bad("truncated protobuf field");
        f.p = buf + pos; f.n = (size_t)n; pos += (size_t)n;
        break;
      }
      case 5: { if (pos + 4 > len) bad("truncated fixed32"); uint32_t v32; memcpy(&v32, buf + pos, 4); f.v = v32; pos += 4; break; }
      default: bad("unsupported protobuf wire type " + std::to_string(f.wt));
    }
    out.push_back(f);
  }
  return out;
}

std::vector<uint8_t> read_file(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) bad(path + ": " + strerror(errno));
  std::vector<uint8_t> d;
  if (fseek(f, 0, SEEK_END) == 0) {
    const long n = ftell(f);
    if (n > 0) { d.resize((size_t)n); rewind(f); if (fread(d.data(), 1, d.size(), f) != d.size()) { fclose(f); bad(path + ": short read"); } }
  }
  fclose(f);
  return d;
}

bool is_file(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }

// contents of the block at (offset, size); checks type and (verify) the masked crc32c of contents + type byte
std::pair<const uint8_t*, size_t> read_block(const std::vector<uint8_t>& d, uint64_t off, uint64_t size, bool verify) {
  if (off > d.size() || size > d.size() - off || kBlockTrailer > d.size() - off - size) bad("block handle points outside the index file");
  const uint8_t* p = d.data() + off;
  uint32_t stored;
  memcpy(&stored, p + size + 1, 4);
  if (verify && unmask_crc(stored) != film_crc32c(0, p, (int64_t)size + 1)) bad("index block at " + std::to_string(off) + ": crc32c mismatch");
  if (p[size] == 1) bad("snappy-compressed index blocks are not supported (TensorFlow writes bundles uncompressed)");
  if (p[size] != 0) bad("unknown block compression type " + std::to_string((int)p[size]));
  return {p, (size_t)size};
}

template <class Fn>
void for_block_entries(const uint8_t* b, size_t len, Fn&& fn) {
  if (len < 4) bad("block too small");
  uint32_t nrestarts;
  memcpy(&nrestarts, b + len - 4, 4);
  if ((uint64_t)nrestarts * 4 + 4 > len) bad("bad restart array");
  const size_t limit = len - 4 - 4 * (size_t)nrestarts;
  size_t pos = 0;
  std::string key;
  while (pos < limit) {
    const uint64_t shared = get_varint(b, limit, pos), unshared = get_varint(b, limit, pos), vlen = get_varint(b, limit, pos);
    if (shared > key.size() || unshared > limit - pos || vlen > limit - pos - unshared) bad("corrupt block entry");
    key.resize((size_t)shared);
    key.append((const char*)b + pos, (size_t)unshared);
    pos += (size_t)unshared;
    fn(key, b + pos, (size_t)vlen);
    pos += (size_t)vlen;
  }
}

struct Entry {
  int dtype = 0, shard = 0;
  std::vector<int64_t> shape;
  uint64_t offset = 0, size = 0;
  uint32_t crc = 0;
  bool slices = false;
};

Entry parse_entry(const uint8_t* p, size_t n) {
  Entry e;
  for (const PbField& f : pb_decode(p, n)) {
    switch (f.field) {
      case 1: e.dtype = (int)f.v; break;
      case 2:
        for (const PbField& d : pb_decode(f.p, f.n))
          if (d.field == 2) {
            int64_t size = 0;
            for (const PbField& s : pb_decode(d.p, d.n)) if (s.field == 1) size = (int64_t)s.v;
            e.shape.push_back(size);
          }
        break;
      case 3: e.shard = (int)f.v; break;
      case 4: e.offset = f.v; break;
      case 5: e.size = f.v; break;
      case 6: e.crc = (uint32_t)f.v; break;
      case 7: e.slices = true; break;
      default: break;
    }
  }
  return e;
}

// object-graph attribute path (without the VARIABLE_VALUE suffix) -> canonical tensor name, "" if it is not a film_net weight.
// The keys come from the file: a hand-written suffix parser over the '/'-separated components (std::regex recurses once per matched
// character - a key with a 100 000-digit run overflowed the stack; round-5 ADVICE).  Keys longer than kMaxKeyLen and numeric
// components longer than 9 digits are not film_net weights.
constexpr size_t kMaxKeyLen = 4096;

bool small_number(const std::string& s, long* out) {
  if (s.empty() || s.size() > 9) return false;
  long v = 0;
  for (char c : s) {
    if (c < '0' || c > '9') return false;
    v = v * 10 + (c - '0');
  }
  *out = v;
  return true;
}

std::string canonical_name(const std::string& path, int specialized_levels) {
  if (path.size() > kMaxKeyLen) return "";
  std::vector<std::string> c;   // components, last first; at most the six the longest pattern needs
  size_t end = path.size();
  while (c.size() < 6) {
    const size_t slash = end == 0 ? std::string::npos : path.rfind('/', end - 1);
    c.push_back(path.substr(slash == std::string::npos ? 0 : slash + 1, end - (slash == std::string::npos ? 0 : slash + 1)));
    if (slash == std::string::npos) break;
    end = slash;
  }
  const size_t n = c.size();
  if (n < 2 || (c[0] != "kernel" && c[0] != "bias")) return "";
  long a = 0, b = 0;
  // .../extract_sublevels/convs/<i>/(kernel|bias)
  if (n >= 4 && c[2] == "convs" && c[3] == "extract_sublevels" && small_number(c[1], &a))
    return "feat_net/sub_extractor/cfeat_conv_" + std::to_string(a) + "/" + c[0];
  // .../_predictors/<p>/_convs/<j>/(kernel|bias)
  if (n >= 5 && c[2] == "_convs" && c[4] == "_predictors" && small_number(c[3], &a) && small_number(c[1], &b))
    return "predict_flow/" + (a < specialized_levels ? "flow_predictor_" + std::to_string(a) : std::string("flow_predictor_shared")) + "/conv_" + std::to_string(b) + "/" + c[0];
  // .../convs/<i>/<j>/(kernel|bias)
  if (n >= 4 && c[3] == "convs" && small_number(c[2], &a) && small_number(c[1], &b))
    return "fusion/convs_" + std::to_string(a) + "_" + std::to_string(b) + "/" + c[0];
  // .../output_conv/(kernel|bias)
  if (c[1] == "output_conv") return "fusion/output_conv/" + c[0];
  return "";
}

// "natural" order of keys (digit runs compare as numbers): the pool of rule 2 is walked in this order, as in tf_bundle.py
bool natural_less(const std::string& a, const std::string& b) {
  size_t i = 0, j = 0;
  while (i < a.size() && j < b.size()) {
    if (isdigit((unsigned char)a[i]) && isdigit((unsigned char)b[j])) {
      size_t i2 = i, j2 = j;
      while (i2 < a.size() && isdigit((unsigned char)a[i2])) ++i2;
      while (j2 < b.size() && isdigit((unsigned char)b[j2])) ++j2;
      auto strip = [](std::string v) { const size_t nz = v.find_first_not_of('0'); return nz == std::string::npos ? std::string("0") : v.substr(nz); };
      const std::string xs = strip(a.substr(i, i2 - i)), ys = strip(b.substr(j, j2 - j));
      if (xs.size() != ys.size()) return xs.size() < ys.size();
      if (xs != ys) return xs < ys;
      i = i2; j = j2;
    } else {
      if (a[i] != b[j]) return a[i] < b[j];
      ++i; ++j;
    }
  }
  return a.size() - i < b.size() - j;
}

std::string shape_str(const std::vector<int64_t>& s) {
  std::string o = "(";
  for (size_t i = 0; i < s.size(); ++i) o += (i ? "," : "") + std::to_string(s[i]);
  return o + ")";
}

struct Bundle {
  std::string prefix;
  int num_shards = 1;
  bool verify = true;
  std::map<std::string, Entry> entries;
  std::map<int, FILE*> shards;
  ~Bundle() { for (auto& kv : shards) if (kv.second) fclose(kv.second); }

  void open(const std::string& pfx, bool vfy) {
    prefix = pfx; verify = vfy;
    const std::vector<uint8_t> d = read_file(prefix + ".index");
    if (d.size() < kFooterLen) bad(prefix + ".index: too small to be a table file");
    const uint8_t* footer = d.data() + d.size() - kFooterLen;
    uint64_t magic;
    memcpy(&magic, footer + 40, 8);
    if (magic != kTableMagic) bad(prefix + ".index: bad table magic (not a TensorFlow bundle index)");
    size_t pos = 0;
    (void)get_varint(footer, 40, pos); (void)get_varint(footer, 40, pos);          // metaindex handle: unused
    const uint64_t io = get_varint(footer, 40, pos), is = get_varint(footer, 40, pos);
    bool first = true, have_header = false;
    const auto idx = read_block(d, io, is, verify);
    for_block_entries(idx.first, idx.second, [&](const std::string&, const uint8_t* hv, size_t hn) {
      size_t p = 0;
      const uint64_t bo = get_varint(hv, hn, p), bs = get_varint(hv, hn, p);
      const auto blk = read_block(d, bo, bs, verify);
      for_block_entries(blk.first, blk.second, [&](const std::string& key, const uint8_t* v, size_t n) {
        if (first) {
          first = false;
          if (!key.empty()) bad(prefix + ".index: missing bundle header entry");
          have_header = true;
          int endianness = 0;
          for (const PbField& f : pb_decode(v, n)) { if (f.field == 1) num_shards = (int)f.v; else if (f.field == 2) endianness = (int)f.v; }
          if (endianness != 0) bad("big-endian bundles are not supported");
          if (num_shards < 1 || num_shards > 99999) bad(prefix + ".index: header names " + std::to_string(num_shards) + " shards");
          return;
        }
        entries[key] = parse_entry(v, n);
      });
    });
    if (!have_header) bad(prefix + ".index: missing bundle header entry");
  }

  std::vector<float> tensor(const std::string& key) {
    const Entry& e = entries.at(key);
    if (e.slices) bad(key + ": sliced (partitioned) variables are not supported");
    if (e.dtype != 1) bad(key + ": dtype " + std::to_string(e.dtype) + " is not DT_FLOAT");
    int64_t n = 1;
    for (int64_t s : e.shape) n *= s;
    if (e.shard < 0 || e.shard >= num_shards) bad(key + ": shard " + std::to_string(e.shard) + " of a bundle with " + std::to_string(num_shards));
    if (n < 0 || e.size != (uint64_t)n * 4) bad(key + ": " + std::to_string(e.size) + " bytes for shape " + shape_str(e.shape));
    FILE*& f = shards[e.shard];
    if (!f) {
      char name[64];
      snprintf(name, sizeof name, ".data-%05d-of-%05d", e.shard, num_shards);
      f = fopen((prefix + name).c_str(), "rb");
      if (!f) bad(prefix + name + ": " + strerror(errno));
    }
    std::vector<float> out((size_t)n);
    if (fseeko(f, (off_t)e.offset, SEEK_SET) != 0 || fread(out.data(), 1, (size_t)e.size, f) != (size_t)e.size)
      bad(key + ": data range outside shard " + std::to_string(e.shard));
    if (verify && unmask_crc(e.crc) != film_crc32c(0, out.data(), (int64_t)e.size)) bad(key + ": tensor crc32c mismatch");
    return out;
  }
};

bool ends_with(const std::string& s, const std::string& t) { return s.size() >= t.size() && s.compare(s.size() - t.size(), t.size(), t) == 0; }

}  // namespace

extern "C" int film_load_bundle(film_t* h, const char* path, int verify_crc, char* report, int64_t report_cap, int64_t* report_needed) {
  if (!h || !path) return fail(h, FILM_ERR_INVALID, "NULL argument");
  try {
    std::string prefix;
    const std::string p(path);
    if (is_file(p + "/variables/variables.index")) prefix = p + "/variables/variables";
    else if (is_file(p + ".index")) prefix = p;
    else if (ends_with(p, ".index") && is_file(p)) prefix = p.substr(0, p.size() - 6);
    else return fail(h, FILM_ERR_NOTFOUND, "%s: no SavedModel variables bundle (variables/variables.index) and no bundle prefix", path);
    Bundle rd;
    rd.open(prefix, verify_crc != 0);

    // what the configuration of this handle needs: name -> shape
    std::map<std::string, std::vector<int64_t>> specs;
    std::vector<std::string> spec_order;
    for (const LayerPack& L : h->layers) {
      specs[L.name + "/kernel"] = {L.kh, L.kw, L.cin, L.cout};
      specs[L.name + "/bias"] = {L.cout};
      spec_order.push_back(L.name + "/kernel");
      spec_order.push_back(L.name + "/bias");
    }
    std::vector<std::string> var_keys;
    for (const auto& kv : rd.entries)
      if (ends_with(kv.first, kVarSuffix) && kv.second.dtype == 1 && kv.first.find("/.OPTIMIZER_SLOT/") == std::string::npos &&
          kv.first.compare(0, 9, "optimizer") != 0)
        var_keys.push_back(kv.first);

    std::map<std::string, std::vector<float>> out;
    std::map<std::string, std::pair<std::string, std::string>> rep;   // name -> (rule, key)
    std::set<std::string> used;
    const size_t suffix = strlen(kVarSuffix);
    for (const std::string& k : var_keys) {                              // rule 1: attribute-path patterns
      const std::string name = canonical_name(k.substr(0, k.size() - suffix), h->cfg.specialized_levels);
      auto sp = specs.find(name);
      if (name.empty() || sp == specs.end() || rd.entries[k].shape != sp->second) continue;
      std::vector<float> t = rd.tensor(k);
      auto prev = out.find(name);
      if (prev != out.end() && prev->second != t) bad(name + ": two different variables map to it (" + rep[name].second + " and " + k + ")");
      out[name] = std::move(t);
      rep[name] = {"path", k};
      used.insert(k);
    }
    std::vector<std::string> missing;
    for (const std::string& n : spec_order) if (!out.count(n)) missing.push_back(n);
    if (!missing.empty()) {                                              // rule 2: unique shape on both sides
      std::vector<std::string> pool;
      for (const std::string& k : var_keys) if (!used.count(k)) pool.push_back(k);
      std::sort(pool.begin(), pool.end(), natural_less);
      std::string ambiguous;
      int n_amb = 0;
      for (const std::string& name : missing) {
        std::vector<std::string> cands;
        for (const std::string& k : pool) if (!used.count(k) && rd.entries[k].shape == specs[name]) cands.push_back(k);
        int rivals = 0;
        for (const std::string& n : missing) if (!out.count(n) && specs[n] == specs[name]) ++rivals;
        if (cands.size() == 1 && rivals == 1) {
          out[name] = rd.tensor(cands[0]);
          rep[name] = {"shape", cands[0]};
          used.insert(cands[0]);
        } else if (!cands.empty()) {
          if (n_amb++ < 6) ambiguous += (ambiguous.empty() ? "" : "; ") + name + " <- one of [" + cands[0] + (cands.size() > 1 ? ", " + cands[1] + (cands.size() > 2 ? ", ..." : "") : "") + "]";
        }
      }
      if (n_amb)
        bad(prefix + ": " + std::to_string(n_amb) + " tensor(s) could not be placed by their object-graph path and their shape is not unique among "
            "the remaining variables - refusing to guess: " + ambiguous);
    }
    std::vector<std::string> still;
    for (const std::string& n : spec_order) if (!out.count(n)) still.push_back(n);
    if (!still.empty())
      return fail(h, FILM_ERR_NOTFOUND, "%s: no variable found for %s%s (%d of %d tensors); keys look like %s", prefix.c_str(), still[0].c_str(),
                  still.size() > 1 ? ", ..." : "", (int)still.size(), (int)specs.size(), var_keys.empty() ? "(none)" : var_keys[0].c_str());

    // every tensor has been read, checked and matched before the first one is handed over; should a hand-over still fail, say that the
    // handle now holds a mix of two weight sets (round-5 ADVICE) instead of passing the inner message on alone
    int n_set = 0;
    for (const std::string& n : spec_order) {
      const std::vector<int64_t>& shp = specs[n];
      const int rc = film_set_weight(h, n.c_str(), out[n].data(), shp.data(), (int)shp.size());
      if (rc != FILM_OK) {
        const std::string inner = film_last_error(h);
        return fail(h, rc, "film_load_bundle: %s (after %d of %d tensors: the weights of this handle are now UNDEFINED - load a complete set before using it)",
                    inner.c_str(), n_set, (int)spec_order.size());
      }
      ++n_set;
    }
    const int rc = film_finalize(h);
    if (rc != FILM_OK) {
      const std::string inner = film_last_error(h);
      return fail(h, rc, "film_load_bundle: %s (all %d tensors were handed over; film_finalize failed: the handle cannot run until it succeeds)", inner.c_str(), n_set);
    }

    // report: one line per tensor, "<name>\t<rule>\t<checkpoint key>\n" (rule = path | shape); a tensor placed by its shape is a
    // (unique-shape) guess, not a name match - callers should say so (the Python wrapper logs a warning)
    std::string text;
    for (const std::string& n : spec_order) text += n + "\t" + rep[n].first + "\t" + rep[n].second + "\n";
    if (report_needed) *report_needed = (int64_t)text.size() + 1;
    if (report && report_cap > 0) {
      const size_t n = std::min((size_t)report_cap - 1, text.size());
      memcpy(report, text.data(), n);
      report[n] = 0;
    }
    return FILM_OK;
  } catch (const Err& e) {
    return fail(h, FILM_ERR_INVALID, "film_load_bundle: %s", e.msg.c_str());
  } catch (const std::exception& e) {
    return fail(h, FILM_ERR_INVALID, "film_load_bundle: %s", e.what());
  }
}
