// film_layers.cpp -- the layer table of film_net (names, HWIO shapes, channel permutations of the concatenated inputs) and the
// packer that turns the HWIO tensors into the kernels' weight layouts (K-major, F(4,3), nested F(4,3) x F(2,3), phase-summed
// 2x2, F(2,3), halo, bf16 splits), in contiguous groups that are packed and uploaded on demand; the C-ABI entry points that
// take / give weights (film_set_weight, film_finalize, film_export_packed / film_import_packed, film_export_layouts) and the
// crc32c helper of the SavedModel reader.  Replaces the variable-restore half of tf.saved_model.load (eval/interpolator.py:148).
#include "film_internal.h"

#include <dlfcn.h>

namespace film_internal {

// ---------------------------------------------------------------------------------------------
// Architecture helpers (mirror frame-interpolation_amd/film_hip/weights.py)
// ---------------------------------------------------------------------------------------------
std::vector<int> feature_channels(const film_config& c) {  // feature_extractor.py:186-193
  std::vector<int> out;
  for (int l = 0; l < c.pyramid_levels; ++l) {
    int ch = 0;
    for (int j = 0; j < c.sub_levels; ++j)
      if (j <= l) ch += c.filters << j;
    out.push_back(ch);
  }
  return out;
}
int slot_offset(const film_config& c, int j) {  // channel offset of sub-pyramid stage j in a feature level
  int o = 0;
  for (int k = 0; k < j; ++k) o += c.filters << k;
  return o;
}
std::vector<int> fusion_filters(const film_config& c) {  // fusion.py:75-79
  std::vector<int> out;
  for (int i = 0; i < c.fusion_pyramid_levels - 1; ++i)
    out.push_back(i < c.specialized_levels ? (c.filters << i) : (c.filters << c.specialized_levels));
  return out;
}
std::string predictor_prefix(const film_config& c, int level) {  // pyramid_flow_estimator.py:109-123
  if (level < c.specialized_levels) return "predict_flow/flow_predictor_" + std::to_string(level);
  return "predict_flow/flow_predictor_shared";
}
int predictor_index(const film_config& c, int level) { return std::min(level, c.specialized_levels); }

int validate_config(film_t* h, const film_config& c) {
  if (c.pyramid_levels < 1 || c.pyramid_levels > 12) return fail(h, FILM_ERR_INVALID, "pyramid_levels out of range");
  if (c.pyramid_levels < c.fusion_pyramid_levels || c.fusion_pyramid_levels < 2)
    return fail(h, FILM_ERR_INVALID, "config.pyramid_levels must be greater than or equal to config.fusion_pyramid_levels.");
  if (c.specialized_levels < 1 || c.specialized_levels > c.pyramid_levels || c.specialized_levels > FILM_MAX_SPECIALIZED)
    return fail(h, FILM_ERR_INVALID, "specialized_levels out of range");
  if (c.sub_levels < 1 || c.sub_levels > c.specialized_levels + 1)
    return fail(h, FILM_ERR_INVALID, "sub_levels must be within [1, specialized_levels+1]");
  if (c.filters <= 0 || c.filters % 32) return fail(h, FILM_ERR_INVALID, "filters must be a positive multiple of 32");
  for (int i = 0; i <= c.specialized_levels; ++i) {
    int nf = c.flow_filters[i];
    if (nf <= 0 || nf % 32 || !(nf / 2 == 16 || (nf / 2) % 32 == 0))
      return fail(h, FILM_ERR_INVALID, "flow_filters[%d]=%d unsupported (need 32 or a multiple of 64)", i, nf);
    if (c.flow_convs[i] < 1) return fail(h, FILM_ERR_INVALID, "flow_convs[%d] must be >= 1", i);
  }
  return FILM_OK;
}

// ---------------------------------------------------------------------------------------------
// Layer table + packing
// ---------------------------------------------------------------------------------------------
std::vector<int> identity_perm(int n) {
  std::vector<int> p(n);
  for (int i = 0; i < n; ++i) p[i] = i;
  return p;
}
// internal channel order of an aligned-pyramid level: [feat0 C | feat1 C | img0 3 | img1 3 | bflow 2 | fflow 2 | 0 x6]
// reference order (interpolator.py:167-183):          [img0 3 | feat0 C | img1 3 | feat1 C | bflow 2 | fflow 2]
std::vector<int> aligned_perm(int C) {
  std::vector<int> p;
  for (int c = 0; c < C; ++c) p.push_back(3 + c);
  for (int c = 0; c < C; ++c) p.push_back(3 + C + 3 + c);
  for (int j = 0; j < 3; ++j) p.push_back(j);
  for (int j = 0; j < 3; ++j) p.push_back(3 + C + j);
  for (int j = 0; j < 4; ++j) p.push_back(2 * (3 + C) + j);
  for (int j = 0; j < 6; ++j) p.push_back(-1);
  return p;
}

void build_layers(film_t* h) {
  const film_config& c = h->cfg;
  h->layers.clear();
  h->layer_idx.clear();
  auto add = [&](const std::string& name, int kh, int kw, int cin, int cout, std::vector<int> perm) {
    LayerPack L;
    L.name = name; L.kh = kh; L.kw = kw; L.cin = cin; L.cout = cout; L.perm = std::move(perm);
    h->layer_idx[name] = (int)h->layers.size();
    h->layers.push_back(std::move(L));
  };
  int cin = 3;
  for (int i = 0; i < c.sub_levels; ++i) {
    int k = c.filters << i;
    add("feat_net/sub_extractor/cfeat_conv_" + std::to_string(2 * i), 3, 3, cin, k, identity_perm(cin));
    if (i == 0) h->layers.back().c3 = true;
    add("feat_net/sub_extractor/cfeat_conv_" + std::to_string(2 * i + 1), 3, 3, k, k, identity_perm(k));
    cin = k;
  }
  auto fc = feature_channels(c);
  for (int p = 0; p <= c.specialized_levels; ++p) {
    std::string prefix = predictor_prefix(c, p);
    int ci = 2 * fc[std::min(p, c.pyramid_levels - 1)];
    int nf = c.flow_filters[p], nconv = c.flow_convs[p];
    for (int j = 0; j < nconv; ++j) {
      add(prefix + "/conv_" + std::to_string(j), 3, 3, ci, nf, identity_perm(ci));
      ci = nf;
    }
    add(prefix + "/conv_" + std::to_string(nconv), 1, 1, nf, nf / 2, identity_perm(nf));
    add(prefix + "/conv_" + std::to_string(nconv + 1), 1, 1, nf / 2, 2, identity_perm(nf / 2));
  }
  auto ff = fusion_filters(c);
  const int FL = c.fusion_pyramid_levels;
  for (int i = 0; i < FL - 1; ++i) {
    const int aligned_ref = 2 * (3 + fc[i]) + 4;
    std::vector<int> p0;
    int net_c;
    if (i == FL - 2) { net_c = 2 * (3 + fc[FL - 1]) + 4; p0 = aligned_perm(fc[FL - 1]); }
    else { net_c = ff[i + 1]; p0 = identity_perm(net_c); }
    add("fusion/convs_" + std::to_string(i) + "_0", 2, 2, net_c, ff[i], p0);
    std::vector<int> p1 = aligned_perm(fc[i]);
    for (int j = 0; j < ff[i]; ++j) p1.push_back(aligned_ref + j);
    add("fusion/convs_" + std::to_string(i) + "_1", 3, 3, aligned_ref + ff[i], ff[i], p1);
    add("fusion/convs_" + std::to_string(i) + "_2", 3, 3, ff[i], ff[i], identity_perm(ff[i]));
  }
  add("fusion/output_conv", 1, 1, ff[0], 3, identity_perm(ff[0]));
  // Weight layouts in four contiguous GROUPS, packed on demand (film_finalize packs group 0; the planner asks for the
  // others when a plan first needs them) so that the default fp32 path neither builds nor broadcasts the copies it never
  // reads:  0 = what the default plan runs on: K-major / first-layer / 1x1 layouts + biases, the phase-summed 2x2
  //             layers, the F(4,3) copy                                                     (3.1x the parameters)
  //         1 = F(2,3) copy (conv_wino_kernel: levels narrower than the F(4,3) patches)
  //         2 = halo copy (conv_halo_kernel: option winograd = 0 / halo_all)
  //         3 = bf16 split copies (precision modes bf16x6 / bf16x3)
  int64_t off = 0;
  auto al = [&]() { off = (off + 3) & ~int64_t(3); };
  for (auto& L : h->layers) {
    L.w_off = off; off += L.packed_rows() * L.cout; al();
    L.b_off = off; off += L.cout; al();
    if (L.has_fold()) { L.wf_off = off; off += (int64_t)9 * L.ctot() * L.cout; al(); }
    if (L.has_fold() && L.ctot() % 16 == 0) { L.wf4_off = off; off += (int64_t)4 * L.ctot() * L.cout; al(); }
    if (L.has_halo()) { L.w43_off = off; off += L.packed_rows() * L.cout / 9 * 18; al(); }
    if (L.has_w2d()) { L.w2d_off = off; off += L.packed_rows() * L.cout / 9 * 24; al(); }
  }
  h->group_end[0] = off;
  for (auto& L : h->layers)
    if (L.has_halo()) { L.ww_off = off; off += L.packed_rows() * L.cout / 9 * 12; al(); }
  h->group_end[1] = off;
  for (auto& L : h->layers)
    if (L.has_halo()) { L.wh_off = off; off += L.packed_rows() * L.cout; al(); }
  h->group_end[2] = off;
  for (auto& L : h->layers) {
    if (L.has_fold()) { L.wfx_off = off; off += (int64_t)9 * L.ctot() * L.cout; al(); }
    if (L.has_halo()) {
      L.ws_off = off; off += (L.packed_rows() * L.cout * 3 + 1) / 2; al();
      L.wx_off = off; off += L.packed_rows() * L.cout / 9 * 12; al();
    }
  }
  h->group_end[3] = off;
  h->packed_floats = 0;
  h->groups_packed = 0;
}


}  // namespace film_internal

using namespace film_internal;

// float -> bfloat16, round to nearest even (what v_cvt_pk_bf16_f32 does; weights are finite)
static inline uint16_t bf16_rne(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf16_to_float(uint16_t b) {
  const uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}


extern "C" {

// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), slicing-by-8.  Host-side helper of the SavedModel
// variables reader (film_hip/tf_bundle.py): TensorFlow stores a masked crc32c per tensor and per index block.
uint32_t film_crc32c(uint32_t crc, const void* data, int64_t n) {
  static uint32_t tab[8][256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int t = 1; t < 8; ++t) tab[t][i] = (tab[t - 1][i] >> 8) ^ tab[0][tab[t - 1][i] & 0xFF];
    init = true;
  }
  const uint8_t* p = static_cast<const uint8_t*>(data);
  uint32_t c = ~crc;
  while (n > 0 && (reinterpret_cast<uintptr_t>(p) & 7)) { c = tab[0][(c ^ *p++) & 0xFF] ^ (c >> 8); --n; }
  while (n >= 8) {
    uint64_t v;
    memcpy(&v, p, 8);
    v ^= c;
    c = tab[7][v & 0xFF] ^ tab[6][(v >> 8) & 0xFF] ^ tab[5][(v >> 16) & 0xFF] ^ tab[4][(v >> 24) & 0xFF] ^
        tab[3][(v >> 32) & 0xFF] ^ tab[2][(v >> 40) & 0xFF] ^ tab[1][(v >> 48) & 0xFF] ^ tab[0][(v >> 56) & 0xFF];
    p += 8; n -= 8;
  }
  while (n-- > 0) c = tab[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return ~c;
}

int film_set_weight(film_t* h, const char* name, const float* data, const int64_t* dims, int ndim) {
  if (!h || !name || !data || !dims || ndim < 1 || ndim > 4) return fail(h, FILM_ERR_INVALID, "bad argument");
  std::string nm(name);
  const size_t slash = nm.rfind('/');
  if (slash == std::string::npos) return fail(h, FILM_ERR_NOTFOUND, "unknown weight '%s'", name);
  const std::string layer = nm.substr(0, slash), kind = nm.substr(slash + 1);
  auto it = h->layer_idx.find(layer);
  if (it == h->layer_idx.end() || (kind != "kernel" && kind != "bias")) return fail(h, FILM_ERR_NOTFOUND, "unknown weight '%s'", name);
  const LayerPack& L = h->layers[it->second];
  int64_t n = 1;
  for (int i = 0; i < ndim; ++i) n *= dims[i];
  if (kind == "kernel") {
    if (ndim != 4 || dims[0] != L.kh || dims[1] != L.kw || dims[2] != L.cin || dims[3] != L.cout)
      return fail(h, FILM_ERR_INVALID, "%s: expected HWIO [%d,%d,%d,%d]", name, L.kh, L.kw, L.cin, L.cout);
  } else if (ndim != 1 || dims[0] != L.cout) {
    return fail(h, FILM_ERR_INVALID, "%s: expected [%d]", name, L.cout);
  }
  HostTensor t;
  t.dims.assign(dims, dims + ndim);
  t.data.assign(data, data + n);
  h->host_w[nm] = std::move(t);
  h->finalized = false;
  return FILM_OK;
}

// Uploads the floats [from, to) of the packed blob.  The device buffer is sized for every layout group once (1.1 GB of
// 288): groups packed later land at their fixed offsets and no plan has to be rebuilt.
static int upload_packed(film_t* h, int64_t from, int64_t to) {
  if (h->plan_only || to <= from) return FILM_OK;
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->packed_dev) HIPCHK(h, hipMalloc(&h->packed_dev, (size_t)h->group_end[3] * sizeof(float)));
  HIPCHK(h, hipMemcpy(h->packed_dev + from, h->packed_host.data() + from, (size_t)(to - from) * sizeof(float), hipMemcpyHostToDevice));
  return FILM_OK;
}

// Packs the layouts of `group` of layer L for the output channels [co0, co1) (the unit of work of the packing threads;
// every output channel owns disjoint ranges of every layout).  Layers without a per-channel layout (first layer, 1x1
// heads) and the biases are handled by the caller of co0 == 0.
static void pack_layer_group(film_t* h, const LayerPack& L, int group, int co0, int co1) {
  const float* src = h->host_w.at(L.name + "/kernel").data.data();
  float* const base = h->packed_host.data();
  const int ct = L.ctot();
  if (group == 0 && co0 == 0) {
    memcpy(base + L.b_off, h->host_w.at(L.name + "/bias").data.data(), sizeof(float) * L.cout);
    float* dst = base + L.w_off;
    if (L.c3) {
      for (int tap = 0; tap < 9; ++tap)
        for (int c = 0; c < 3; ++c)
          memcpy(dst + ((size_t)tap * 4 + c) * L.cout, src + ((size_t)tap * 3 + c) * L.cout, sizeof(float) * L.cout);
    } else if (!L.kmajor()) {
      for (int tap = 0; tap < L.kh * L.kw; ++tap)
        for (int ci = 0; ci < ct; ++ci) {
          const int ref = L.perm[ci];
          if (ref < 0) continue;  // zero row (padding channel)
          memcpy(dst + ((size_t)tap * ct + ci) * L.cout, src + ((size_t)tap * L.cin + ref) * L.cout, sizeof(float) * L.cout);
        }
    }
  }
  if (!L.kmajor()) return;
  const int ntap = L.kh * L.kw;
  const size_t ktot = (size_t)ntap * ct;
  const size_t nkc = (size_t)ct / 16;
  // ---- K-major copy (group 0), halo copy (group 2), bf16x6 planes (group 3): one pass per (tap, 16-channel chunk); the
  // 16 source rows (each `cout` contiguous floats) stay in L1 while every output channel receives its 16 k values
  float* dk = group == 0 ? base + L.w_off : nullptr;
  float* dh = group == 2 && L.wh_off >= 0 ? base + L.wh_off : nullptr;
  uint16_t* ds = group == 3 && L.ws_off >= 0 ? reinterpret_cast<uint16_t*>(base + L.ws_off) : nullptr;
  if (dk || dh || ds)
    for (int tap = 0; tap < ntap; ++tap)
      for (size_t kc = 0; kc < nkc; ++kc) {
        const float* rows[16];
        for (int j = 0; j < 16; ++j) {
          const int ref = L.perm[kc * 16 + j];
          rows[j] = ref < 0 ? nullptr : src + ((size_t)tap * L.cin + ref) * L.cout;  // nullptr: zero (padding) channel
        }
        for (int co = co0; co < co1; ++co) {
          float v[16];
          for (int j = 0; j < 16; ++j) v[j] = rows[j] ? rows[j][co] : 0.f;
          if (dk) memcpy(dk + (size_t)co * ktot + (size_t)tap * ct + kc * 16, v, sizeof(v));
          if (dh) memcpy(dh + (((size_t)co * nkc + kc) * 9 + tap) * 16, v, sizeof(v));
          if (ds) {  // exact 3-way bf16 split, round-to-nearest-even pieces (same as conv_split4 on the device)
            uint16_t* d = ds + (((size_t)co * nkc + kc) * 9 + tap) * 48;
            for (int j = 0; j < 16; ++j) {
              const uint16_t hb = bf16_rne(v[j]);
              const float r = v[j] - bf16_to_float(hb);
              const uint16_t mb = bf16_rne(r);
              const float q = r - bf16_to_float(mb);
              d[j] = hb; d[16 + j] = mb; d[32 + j] = bf16_rne(q);
            }
          }
        }
      }
  // ---- sub-pixel phases of upsample + 2x2: weights of the taps that read the same input pixel, summed (fp32: group 0;
  // bf16 hi / mid for conv_foldx3_kernel: group 3)
  if (L.wf_off >= 0 && (group == 0 || group == 3)) {
    float* df = base + L.wf_off;
    uint16_t* dfx = reinterpret_cast<uint16_t*>(base + L.wfx_off);
    const size_t nk16f = (size_t)ct / 16;
    // step of (tap a*2+b, phase py*2+px) in conv_foldx3_kernel's order (taps 00 00 00 | 00 01 01 | 10 10 11)
    static const int kFoldStep[4][4] = {{0, 1, 2, 3}, {-1, 4, -1, 5}, {-1, -1, 6, 7}, {-1, -1, -1, 8}};
    for (int py = 0; py < 2; ++py)
      for (int px = 0; px < 2; ++px) {
        const int nt = (py + 1) * (px + 1);
        const size_t kph = (size_t)nt * ct;
        int t = 0;
        for (int a = 0; a <= py; ++a)
          for (int b = 0; b <= px; ++b, ++t)
            for (int ci = 0; ci < ct; ++ci) {
              const int ref = L.perm[ci];
              if (ref < 0) continue;
              for (int co = co0; co < co1; ++co) {
                float acc = 0.f;  // kernel taps (dy, dx) with (py & dy) == a and (px & dx) == b, in raster order
                for (int dy = 0; dy < 2; ++dy)
                  for (int dx = 0; dx < 2; ++dx)
                    if ((py & dy) == a && (px & dx) == b) acc += src[((size_t)(dy * 2 + dx) * L.cin + ref) * L.cout + co];
                if (group == 0) df[(size_t)co * kph + (size_t)t * ct + ci] = acc;
                else {
                  const int step = kFoldStep[a * 2 + b][py * 2 + px];
                  uint16_t* d = dfx + (((size_t)co * nk16f + ci / 16) * 9 + step) * 32 + ci % 16;
                  const uint16_t hb = bf16_rne(acc);
                  d[0] = hb;
                  d[16] = bf16_rne(acc - bf16_to_float(hb));
                }
              }
            }
        df += kph * L.cout;
      }
  }
  // ---- difference form of upsample + 2x2 (group 0; conv_fold4_impl.h): [Cout/32][chunk8][plane 4][K half][32][4], planes S = ((W00 + W01) +
  // W10) + W11, Sx = W01 + W11, Sy = W10 + W11, W11 (padding channels: zero rows)
  if (group == 0 && L.wf4_off >= 0) {
    float* d4 = base + L.wf4_off;
    const size_t nk8 = (size_t)ct / 8;
    for (int ci = 0; ci < ct; ++ci) {
      const int ref = L.perm[ci];
      for (int co = co0; co < co1; ++co) {
        float w[4] = {0.f, 0.f, 0.f, 0.f};
        if (ref >= 0)
          for (int tp = 0; tp < 4; ++tp) w[tp] = src[((size_t)tp * L.cin + ref) * L.cout + co];
        const float pl[4] = {((w[0] + w[1]) + w[2]) + w[3], w[1] + w[3], w[2] + w[3], w[3]};
        for (int q = 0; q < 4; ++q)
          d4[((((size_t)(co / 32) * nk8 + ci / 8) * 4 + q) * 2 + (ci % 8) / 4) * 128 + (co % 32) * 4 + ci % 4] = pl[q];
      }
    }
  }
  // ---- nested Winograd copy (group 0, deep-K layers): U[mu][nu] = the F(2,3) transform along dy of the F(4,3)-transformed
  // kernel rows u_nu(dy) (the same u as the w43 copy), [Cout/32][chunk8][mu 4][nu 6][K half][32][4]
  if (group == 0 && L.w2d_off >= 0) {
    float* d2 = base + L.w2d_off;
    const size_t nk8 = (size_t)ct / 8;
    for (size_t kc = 0; kc < nk8; ++kc) {
      const float* rows[3][3][8];
      for (int dy = 0; dy < 3; ++dy)
        for (int dx = 0; dx < 3; ++dx)
          for (int j = 0; j < 8; ++j) {
            const int ref = L.perm[kc * 8 + j];
            rows[dy][dx][j] = ref < 0 ? nullptr : src + ((size_t)(dy * 3 + dx) * L.cin + ref) * L.cout;
          }
      for (int co = co0; co < co1; ++co)
        for (int j = 0; j < 8; ++j) {
          float u[3][6];
          for (int dy = 0; dy < 3; ++dy) {
            const float g0 = rows[dy][0][j] ? rows[dy][0][j][co] : 0.f, g1 = rows[dy][1][j] ? rows[dy][1][j][co] : 0.f,
                        g2 = rows[dy][2][j] ? rows[dy][2][j][co] : 0.f;
            const float e = g0 * (1.f / 24.f) + g2 * (1.f / 6.f), o = g1 * (1.f / 12.f);
            u[dy][0] = g0 * 0.25f;
            u[dy][1] = -((g0 + g2) + g1) * (1.f / 6.f);
            u[dy][2] = -((g0 + g2) - g1) * (1.f / 6.f);
            u[dy][3] = e + o;
            u[dy][4] = e - o;
            u[dy][5] = g2;
          }
          for (int nu = 0; nu < 6; ++nu) {
            const float U[4] = {u[0][nu], ((u[0][nu] + u[2][nu]) + u[1][nu]) * 0.5f, ((u[0][nu] + u[2][nu]) - u[1][nu]) * 0.5f, u[2][nu]};
            for (int mu = 0; mu < 4; ++mu)
              d2[(((((size_t)(co / 32) * nk8 + kc) * 4 + mu) * 6 + nu) * 2 + j / 4) * 128 + (co % 32) * 4 + j % 4] = U[mu];
          }
        }
    }
  }
  // ---- Winograd copies along x: F(4,3) [Cout][chunk8][dy][nu 6][8] (group 0), F(2,3) [Cout][chunk8][nu*3+dy][8]
  // (u0 = g0, u1 = ((g0+g2)+g1)/2, u2 = ((g0+g2)-g1)/2, u3 = g2; group 1) and its bf16 hi / mid planes (group 3)
  if (L.ww_off >= 0 && (group == 0 || group == 1 || group == 3)) {
    float* dw = base + L.ww_off;
    uint16_t* dx3 = reinterpret_cast<uint16_t*>(base + L.wx_off);
    float* d43 = base + L.w43_off;
    const size_t nk8 = (size_t)ct / 8, nk16 = (size_t)ct / 16;
    for (int dy = 0; dy < 3; ++dy)
      for (size_t kc = 0; kc < nk8; ++kc) {
        const float* rows[3][8];
        for (int dx = 0; dx < 3; ++dx)
          for (int j = 0; j < 8; ++j) {
            const int ref = L.perm[kc * 8 + j];
            rows[dx][j] = ref < 0 ? nullptr : src + ((size_t)(dy * 3 + dx) * L.cin + ref) * L.cout;
          }
        for (int co = co0; co < co1; ++co) {
          float g[3][8];
          for (int dx = 0; dx < 3; ++dx)
            for (int j = 0; j < 8; ++j) g[dx][j] = rows[dx][j] ? rows[dx][j][co] : 0.f;
          if (group == 0) {
            float* d = d43 + ((((size_t)co * nk8 + kc) * 3 + dy) * 6) * 8;
            for (int j = 0; j < 8; ++j) {
              const float g0 = g[0][j], g1 = g[1][j], g2 = g[2][j];
              const float e = g0 * (1.f / 24.f) + g2 * (1.f / 6.f), o = g1 * (1.f / 12.f);
              d[0 * 8 + j] = g0 * 0.25f;
              d[1 * 8 + j] = -((g0 + g2) + g1) * (1.f / 6.f);
              d[2 * 8 + j] = -((g0 + g2) - g1) * (1.f / 6.f);
              d[3 * 8 + j] = e + o;
              d[4 * 8 + j] = e - o;
              d[5 * 8 + j] = g2;
            }
            continue;
          }
          float u[4][8];
          for (int j = 0; j < 8; ++j) {
            const float g0 = g[0][j], g1 = g[1][j], g2 = g[2][j];
            u[0][j] = g0; u[1][j] = ((g0 + g2) + g1) * 0.5f; u[2][j] = ((g0 + g2) - g1) * 0.5f; u[3][j] = g2;
          }
          for (int nu = 0; nu < 4; ++nu) {
            if (group == 1) { memcpy(dw + (((size_t)co * nk8 + kc) * 12 + nu * 3 + dy) * 8, u[nu], sizeof(u[nu])); continue; }
            // the same transformed weights as nearest bf16 hi / mid planes (nu = 2h + j)
            uint16_t* d = dx3 + ((((size_t)co * nk16 + kc / 2) * 3 + dy) * 2 + (nu & 1)) * 64 + (nu >> 1) * 32 + (kc & 1) * 8;
            for (int j = 0; j < 8; ++j) {
              const uint16_t hb = bf16_rne(u[nu][j]);
              d[j] = hb;
              d[16 + j] = bf16_rne(u[nu][j] - bf16_to_float(hb));
            }
          }
        }
      }
  }
}

// Packs layout groups [h->groups_packed, n) from the HWIO tensors (kept on the host) and uploads them.  Work items =
// (layer, 32 output channels), pulled from an atomic counter by up to 32 threads: 137.7 MB of parameters into the
// default group 0 in well under a second on the hosts this runs on (it took 7 s single-threaded for every layout).
int film_ensure_groups_(film_t* h, int n) {
  if (n <= h->groups_packed) return FILM_OK;
  if (n > 4) n = 4;
  for (const LayerPack& L : h->layers)
    if (!h->host_w.count(L.name + "/kernel") || !h->host_w.count(L.name + "/bias")) return fail(h, FILM_ERR_STATE, "missing weight '%s'", L.name.c_str());
  const int64_t from = h->groups_packed ? h->group_end[h->groups_packed - 1] : 0, to = h->group_end[n - 1];
  h->packed_host.resize((size_t)to, 0.f);
  struct Item { const LayerPack* L; int co0, co1; };
  std::vector<Item> items;
  for (const LayerPack& L : h->layers)
    for (int co = 0; co < L.cout; co += 32) items.push_back({&L, co, std::min(L.cout, co + 32)});
  for (int g = h->groups_packed; g < n; ++g) {
    std::atomic<size_t> next{0};
    auto worker = [&]() {
      for (size_t i; (i = next.fetch_add(1)) < items.size();) pack_layer_group(h, *items[i].L, g, items[i].co0, items[i].co1);
    };
    const unsigned nth = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nth; ++t) pool.emplace_back(worker);
    worker();
    for (auto& t : pool) t.join();
  }
  h->groups_packed = n;
  h->packed_floats = to;
  return upload_packed(h, from, to);
}

// layout groups the current options need at once (the planner asks for more when a plan needs them)
static int groups_for_options(const film_t* h) {
  int n = 1;
  if (h->opt_wino == 2) n = std::max(n, 2);
  if (h->opt_wino == 0 || h->opt_halo_all) n = std::max(n, 3);
  if (h->opt_precision) n = 4;
  return n;
}

int film_finalize(film_t* h) {
  if (!h) return FILM_ERR_INVALID;
  // A handle that has already run may hold cached plans whose ops point into layout groups 1..3 (F(2,3), halo, bf16
  // copies pulled in by Planner::need_groups).  A second weight set must reach those regions too, or such a plan
  // would mix the new group-0 layouts with the previous set's copies: re-pack everything that was packed before.
  const int prev = h->groups_packed;
  h->groups_packed = 0;
  h->packed_floats = 0;
  h->packed_host.clear();
  h->finalized = false;
  if (!h->plan_only && prev > 0) {   // replays of the previous weight set may still be in flight on the caller's stream
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
  }
  int rc = film_ensure_groups_(h, std::max(groups_for_options(h), prev));
  if (rc) return rc;
  h->finalized = true;
  return FILM_OK;
}

// ---- the parameter set as ONE flat blob (what ranks exchange): per layer, in layer order, the HWIO kernel then the bias ----
static int64_t flat_floats(const film_t* h) {
  int64_t n = 0;
  for (const LayerPack& L : h->layers) n += (int64_t)L.kh * L.kw * L.cin * L.cout + L.cout;
  return n;
}

int film_packed_size(film_t* h, int64_t* n) {
  if (!h || !n) return FILM_ERR_INVALID;
  *n = flat_floats(h);
  return FILM_OK;
}

int film_export_packed(film_t* h, float* dst, int64_t cap, int mem_kind) {
  if (!h || !dst) return FILM_ERR_INVALID;
  if (!h->finalized) return fail(h, FILM_ERR_STATE, "film_finalize has not been called");
  const int64_t n = flat_floats(h);
  if (cap < n) return fail(h, FILM_ERR_INVALID, "capacity %lld < %lld floats", (long long)cap, (long long)n);
  std::vector<float> tmp;
  float* out = dst;
  if (mem_kind != FILM_MEM_HOST) {
    if (h->plan_only) return fail(h, FILM_ERR_NO_DEVICE, "plan-only handle has no device");
    tmp.resize((size_t)n);
    out = tmp.data();
  }
  int64_t off = 0;
  for (const LayerPack& L : h->layers) {
    const HostTensor& k = h->host_w.at(L.name + "/kernel");
    const HostTensor& bq = h->host_w.at(L.name + "/bias");
    memcpy(out + off, k.data.data(), k.data.size() * sizeof(float)); off += (int64_t)k.data.size();
    memcpy(out + off, bq.data.data(), bq.data.size() * sizeof(float)); off += (int64_t)bq.data.size();
  }
  if (mem_kind != FILM_MEM_HOST) {
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy(dst, tmp.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  }
  return FILM_OK;
}

int film_import_packed(film_t* h, const float* src, int64_t n, int mem_kind) {
  if (!h || !src) return FILM_ERR_INVALID;
  if (n != flat_floats(h)) return fail(h, FILM_ERR_INVALID, "blob has %lld floats, expected %lld", (long long)n, (long long)flat_floats(h));
  std::vector<float> tmp;
  const float* in = src;
  if (mem_kind != FILM_MEM_HOST) {
    if (h->plan_only) return fail(h, FILM_ERR_NO_DEVICE, "plan-only handle has no device");
    tmp.resize((size_t)n);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy(tmp.data(), src, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    in = tmp.data();
  }
  int64_t off = 0;
  for (const LayerPack& L : h->layers) {
    HostTensor k, bq;
    k.dims = {L.kh, L.kw, L.cin, L.cout};
    k.data.assign(in + off, in + off + (int64_t)L.kh * L.kw * L.cin * L.cout); off += (int64_t)k.data.size();
    bq.dims = {L.cout};
    bq.data.assign(in + off, in + off + L.cout); off += L.cout;
    h->host_w[L.name + "/kernel"] = std::move(k);
    h->host_w[L.name + "/bias"] = std::move(bq);
  }
  return film_finalize(h);
}

// ---- RCCL weight broadcast (include/film_hip.h).  ncclBroadcast is looked up at run time: a process has ONE RCCL (PyTorch bundles its own),
// and a host that brings a communicator has it loaded already.
namespace {
typedef int (*nccl_bcast_fn)(const void*, void*, size_t, int /* ncclDataType_t */, int, void* /* ncclComm_t */, hipStream_t);
nccl_bcast_fn resolve_nccl_broadcast(std::string* where) {
  if (void* f = dlsym(RTLD_DEFAULT, "ncclBroadcast")) { *where = "the process"; return reinterpret_cast<nccl_bcast_fn>(f); }
  for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
    if (void* lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL))
      if (void* f = dlsym(lib, "ncclBroadcast")) { *where = name; return reinterpret_cast<nccl_bcast_fn>(f); }
  }
  return nullptr;
}
}  // namespace

int film_bcast_weights(film_t* h, void* nccl_comm, int root, int rank, void* stream) {
  if (!h || !nccl_comm || root < 0 || rank < 0) return fail(h, FILM_ERR_INVALID, "bad argument");
  if (h->plan_only) return fail(h, FILM_ERR_NO_DEVICE, "plan-only handle: the RCCL broadcast needs a HIP device");
  if (rank == root && !h->finalized) return fail(h, FILM_ERR_STATE, "the root rank must hold a finalized weight set (film_finalize / film_load_bundle)");
  static std::string where;
  static const nccl_bcast_fn bcast = resolve_nccl_broadcast(&where);
  if (!bcast) return fail(h, FILM_ERR_NOTFOUND, "ncclBroadcast not found: no RCCL in the process and librccl.so cannot be loaded");
  const int64_t n = flat_floats(h);
  HIPCHK(h, hipSetDevice(h->device));
  float* blob = nullptr;
  HIPCHK(h, hipMalloc(&blob, (size_t)n * sizeof(float)));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  int rc = FILM_OK;
  if (rank == root) rc = film_export_packed(h, blob, n, FILM_MEM_DEVICE);
  if (rc == FILM_OK) {
    const int nrc = bcast(blob, blob, (size_t)n, 7 /* ncclFloat32 */, root, nccl_comm, s);
    if (nrc != 0) rc = fail(h, FILM_ERR_HIP, "ncclBroadcast (from %s) failed with ncclResult_t %d", where.c_str(), nrc);
  }
  if (rc == FILM_OK && hipStreamSynchronize(s) != hipSuccess) rc = fail(h, FILM_ERR_HIP, "hipStreamSynchronize behind the broadcast failed: %s", hipGetErrorString(hipGetLastError()));
  if (rc == FILM_OK && rank != root) rc = film_import_packed(h, blob, n, FILM_MEM_DEVICE);
  (void)hipFree(blob);
  return rc;
}

// the kernel-layout blob (debug / tests): the packed prefix [0, *n)
int film_export_layouts(film_t* h, float* dst, int64_t cap, int64_t* n) {
  if (!h) return FILM_ERR_INVALID;
  if (!h->finalized) return fail(h, FILM_ERR_STATE, "film_finalize has not been called");
  if (n) *n = h->packed_floats;
  if (!dst) return FILM_OK;
  if (cap < h->packed_floats) return fail(h, FILM_ERR_INVALID, "capacity %lld < %lld floats", (long long)cap, (long long)h->packed_floats);
  memcpy(dst, h->packed_host.data(), (size_t)h->packed_floats * sizeof(float));
  return FILM_OK;
}


}  // extern "C"
