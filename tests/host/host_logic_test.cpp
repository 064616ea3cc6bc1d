// host_logic_test.cpp -- CPU-only checks of the host logic that needs no GPU: the model-file reader
// and the record layout tables.  Built and run by tests/test_host_logic.py with g++.
#define __host__
#define __device__
#include <cstdio>
#include <cstdlib>
#include <set>
#include <string>

#include <map>
#include <random>

#include "../../target_estimation_amd/csrc/id_table.hpp"
#include "../../target_estimation_amd/csrc/te_layout.hpp"
#include "../../target_estimation_amd/csrc/yaml_mini.hpp"

using namespace te;

static int failures = 0;
#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); ++failures; } \
  } while (0)

// every word of every lane record is used exactly once by (P entries, x, unwrap), and the byte
// offsets of a tile are a bijection onto [0, payload)
template <class M, typename T, int G, int LAYOUT>
void check_layout(const char* name) {
  using C = Cfg<M, T, G, LAYOUT>;
  std::set<long> bytes;
  for (int lane = 0; lane < C::LPT; ++lane)
    for (int w = 0; w < C::RW; ++w) {
      const long off = record_word_offset<C, T>(lane, w);
      CHECK(off >= 0 && off + (long)sizeof(T) <= C::TILE_PAYLOAD);
      CHECK(off % (long)sizeof(T) == 0);
      CHECK(bytes.insert(off).second);
    }
  CHECK((long)bytes.size() * (long)sizeof(T) == C::TILE_PAYLOAD);
  CHECK(C::TILE_BYTES % 128 == 0 && C::TILE_BYTES >= C::TILE_PAYLOAD);
  // P words: each stored (r, c) maps into [0, PW); full layout is injective per lane
  std::set<int> words[64];
  int stored = 0;
  for (int r = 0; r < C::N; ++r)
    for (int c = 0; c < C::N; ++c) {
      const int w = C::p_word(r, c);
      CHECK(w == C::PWORD.v[r][c]);
      if (w < 0) { CHECK(C::SEP && group_of(M::TYPE, r) != group_of(M::TYPE, c)); continue; }
      if (C::TRI_FOLD && w >= C::PW) {   // the remainder of a packed triangle rides in an unwrap slot that no angle uses
        CHECK(w >= C::UW_OFF && w < C::UW_OFF + C::UW && !C::uw_slot_used(C::p_lane(r, c), w - C::UW_OFF));
      } else {
        CHECK(w < C::PW);
      }
      ++stored;
      if (LAYOUT == LAYOUT_PACKED || LAYOUT == LAYOUT_SEPARABLE_PACKED) {
        CHECK(w == C::p_word(c, r) && C::p_lane(r, c) == C::p_lane(c, r) && C::p_lane(r, c) >= 0 && C::p_lane(r, c) < G);
        continue;
      }
      CHECK(words[r % G].insert(w).second);
    }
  if (LAYOUT == LAYOUT_FULL) CHECK(stored == C::N * C::N);
  if (LAYOUT == LAYOUT_PACKED) {   // the G lanes of a target hold the upper triangle exactly once, in slices of PW words
    std::set<std::pair<int, int>> cells;
    for (int r = 0; r < C::N; ++r)
      for (int c = r; c < C::N; ++c) CHECK(cells.insert({C::p_lane(r, c), C::p_word(r, c)}).second);
    CHECK((int)cells.size() == C::TRI);
    if (C::TRI_FOLD) CHECK(C::PW * G + C::TRI_REM == C::TRI && C::RW * G == C::TRI + C::N + 3);   // nothing but the triangle, x and the unwrap memory
    else CHECK(C::PW * G >= C::TRI && C::PW * G < C::TRI + G);
  }
  if (LAYOUT == LAYOUT_SEPARABLE) CHECK(stored == C::PW);
  if (LAYOUT == LAYOUT_SEPARABLE_PACKED) CHECK(stored == 2 * C::PW - C::N);
  CHECK(C::X_OFF == C::PW && C::UW_OFF == C::PW + C::RPL && C::RW == C::PW + C::RPL + C::UW);
  std::printf("layout %-28s RW %3d tile %6ld B (%5.1f B/target)\n", name, C::RW, C::TILE_BYTES, (double)C::TILE_BYTES / C::TPW);
}

// the manager's id table against std::map under a random mix of insert / overwrite / erase / lookup,
// with ids that collide in the table (multiples of large powers of two) and dense ranges
static void check_id_table() {
  std::mt19937 g(5);
  te::IdTable t;
  std::map<unsigned, te::TargetLoc> ref;
  auto pick = [&]() -> unsigned {
    switch (g() % 4) {
      case 0: return g() % 512;                         // dense, many repeats
      case 1: return (g() % 256) << 20;                 // same low bits
      case 2: return 0xFFFFFFFFu - (g() % 64);          // top of the range
      default: return g();
    }
  };
  for (int it = 0; it < 400000; ++it) {
    const unsigned id = pick();
    const int op = g() % 8;
    if (op < 4) {
      const te::TargetLoc loc{(int)(g() % 3), (int)(g() % 100000)};
      t.set(id, loc); ref[id] = loc;
    } else if (op < 6) {
      const bool a = t.erase(id), b = ref.erase(id) != 0;
      CHECK(a == b);
    } else {
      te::TargetLoc loc{-7, -7};
      const bool a = t.find(id, loc);
      auto f = ref.find(id);
      CHECK(a == (f != ref.end()));
      CHECK(t.contains(id) == a);
      if (a && f != ref.end()) { CHECK(loc.batch == f->second.batch && loc.slot == f->second.slot); }
    }
    if (it % 50000 == 0) {
      CHECK(t.size() == ref.size());
      const std::vector<unsigned> ids = t.sorted_ids();
      CHECK(ids.size() == ref.size());
      size_t k = 0;
      for (auto const& kv : ref) { CHECK(k < ids.size() && ids[k] == kv.first); ++k; }   // ascending, as std::map
    }
  }
  te::IdTable big;
  big.reserve(1000000);
  for (unsigned i = 0; i < 1000000; ++i) big.set(i * 7u + 3u, te::TargetLoc{0, (int)i});
  te::TargetLoc loc;
  CHECK(big.size() == 1000000 && big.find(7u * 999999u + 3u, loc) && loc.slot == 999999 && !big.contains(4u));
  std::printf("id table ok (%zu live ids after the random schedule)\n", t.size());
}

// parameter-class rows (te_layout.hpp qr_*): every entry the kernels read has exactly one word, the separable rows hold
// only in-group entries, contiguously
template <class M>
static void check_qr_rows(const char* name) {
  for (int sep = 0; sep < 2; ++sep) {
    const int n = M::N, m = M::K, words = qr_words(M::TYPE, sep != 0);
    std::vector<int> hits((size_t)words, 0);
    int in_group_q = 0, in_group_r = 0;
    for (int r = 0; r < n; ++r)
      for (int c = 0; c < n; ++c) {
        const int w = qr_q_word(M::TYPE, sep != 0, r, c);
        const bool same = group_of(M::TYPE, r) == group_of(M::TYPE, c);
        if (sep && !same) { CHECK(w == -1); continue; }
        CHECK(w >= 0 && w < words);
        ++hits[(size_t)w];
        in_group_q += same ? 1 : 0;
      }
    for (int r = 0; r < m; ++r)
      for (int c = 0; c < m; ++c) {
        const int w = qr_r_word(M::TYPE, sep != 0, r, c);
        const bool same = group_of(M::TYPE, r) == group_of(M::TYPE, c);
        if (sep && !same) { CHECK(w == -1); continue; }
        CHECK(w >= 0 && w < words);
        ++hits[(size_t)w];
        in_group_r += same ? 1 : 0;
      }
    for (int w = 0; w < words; ++w) CHECK(hits[(size_t)w] == 1);
    if (sep) CHECK(words == in_group_q + in_group_r);
    else CHECK(words == n * n + m * m);
  }
  std::printf("qr rows ok: %s (%d dense words, %d separable)\n", name, qr_words(M::TYPE, false), qr_words(M::TYPE, true));
}

int main(int argc, char** argv) {
  check_id_table();
  check_qr_rows<ModelUV>("UV");
  check_qr_rows<ModelUA>("UA");
  check_qr_rows<ModelAV>("AV");
  check_qr_rows<ModelAR>("AR");
  CHECK(qr_words(UNIFORM_VELOCITY, true) == 15 && qr_words(UNIFORM_ACCELERATION, true) == 30);
  CHECK(qr_words(ANGULAR_VELOCITIES, true) == 60 && qr_words(ANGULAR_RATES, true) == 60);
  // a chain's Q block and its R entry are one contiguous run (group-major order)
  CHECK(qr_q_word(ANGULAR_RATES, true, 0, 0) == 0 && qr_q_word(ANGULAR_RATES, true, 12, 12) == 8 && qr_r_word(ANGULAR_RATES, true, 0, 0) == 9);
  CHECK(qr_q_word(ANGULAR_RATES, true, 1, 1) == 10 && qr_r_word(ANGULAR_RATES, true, 5, 5) == 59);
  CHECK(qr_q_word(ANGULAR_VELOCITIES, true, 3, 3) == 15 && qr_r_word(ANGULAR_VELOCITIES, true, 3, 3) == 51 && qr_r_word(ANGULAR_VELOCITIES, true, 5, 5) == 59);
  CHECK((Cfg<ModelAR, double, 1, LAYOUT_SEPARABLE_PACKED>::QR_WORDS == 60) && (Cfg<ModelAR, double, 6, LAYOUT_FULL>::QR_WORDS == 360));
  check_layout<ModelUV, double, 1, LAYOUT_FULL>("UV f64 G1 full");
  check_layout<ModelUV, double, 3, LAYOUT_FULL>("UV f64 G3 full");
  check_layout<ModelUV, float, 3, LAYOUT_FULL>("UV f32 G3 full");
  check_layout<ModelUV, float, 1, LAYOUT_PACKED>("UV f32 G1 packed");
  check_layout<ModelUV, double, 1, LAYOUT_SEPARABLE>("UV f64 separable");
  check_layout<ModelUV, double, 3, LAYOUT_PACKED>("UV f64 G3 packed");
  check_layout<ModelUA, float, 3, LAYOUT_PACKED>("UA f32 G3 packed");
  check_layout<ModelAV, double, 3, LAYOUT_PACKED>("AV f64 G3 packed");
  check_layout<ModelAV, float, 6, LAYOUT_PACKED>("AV f32 G6 packed");
  check_layout<ModelAR, float, 2, LAYOUT_PACKED>("AR f32 G2 packed");
  check_layout<ModelAR, float, 3, LAYOUT_PACKED>("AR f32 G3 packed");
  check_layout<ModelAR, double, 6, LAYOUT_PACKED>("AR f64 G6 packed");
  check_layout<ModelUA, float, 1, LAYOUT_FULL>("UA f32 G1 full");
  check_layout<ModelUA, double, 3, LAYOUT_FULL>("UA f64 G3 full");
  check_layout<ModelUA, double, 1, LAYOUT_PACKED>("UA f64 G1 packed");
  check_layout<ModelUA, float, 1, LAYOUT_SEPARABLE>("UA f32 separable");
  check_layout<ModelAV, double, 3, LAYOUT_FULL>("AV f64 G3 full");
  check_layout<ModelAV, float, 6, LAYOUT_FULL>("AV f32 G6 full");
  check_layout<ModelAV, float, 1, LAYOUT_PACKED>("AV f32 G1 packed");
  check_layout<ModelAV, double, 1, LAYOUT_SEPARABLE>("AV f64 separable");
  check_layout<ModelAR, float, 2, LAYOUT_FULL>("AR f32 G2 full");
  check_layout<ModelAR, double, 3, LAYOUT_FULL>("AR f64 G3 full");
  check_layout<ModelAR, float, 6, LAYOUT_FULL>("AR f32 G6 full");
  check_layout<ModelAR, float, 1, LAYOUT_SEPARABLE>("AR f32 separable");
  check_layout<ModelAR, float, 1, LAYOUT_SEPARABLE_PACKED>("AR f32 separable packed");
  check_layout<ModelAV, double, 1, LAYOUT_SEPARABLE_PACKED>("AV f64 separable packed");
  check_layout<ModelUV, float, 1, LAYOUT_SEPARABLE_PACKED>("UV f32 separable packed");
  check_layout<ModelUA, double, 1, LAYOUT_SEPARABLE_PACKED>("UA f64 separable packed");
  // separable word counts: UV 3*4, UA 3*9, AR 6*9, AV 3*4 + 36
  CHECK((Cfg<ModelUV, float, 1, LAYOUT_SEPARABLE>::PW == 12));
  CHECK((Cfg<ModelUA, float, 1, LAYOUT_SEPARABLE>::PW == 27));
  CHECK((Cfg<ModelAR, float, 1, LAYOUT_SEPARABLE>::PW == 54));
  CHECK((Cfg<ModelAV, float, 1, LAYOUT_SEPARABLE>::PW == 48));

  // model-file reader (reference: src/target_manager.cpp:18-104)
  for (int i = 1; i < argc; ++i) {
    ModelFile mf;
    std::string err;
    CHECK(load_model_file(argv[i], mf, err));
    CHECK(mf.has_frequency && mf.frequency == 250.0);
    const int type = mf.type == "angular_rates" ? 0 : mf.type == "angular_velocities" ? 1 : mf.type == "uniform_acceleration" ? 2 : mf.type == "uniform_velocity" ? 3 : -1;
    CHECK(type >= 0);
    const size_t n = (size_t)model_n(type), m = (size_t)model_m(type);
    CHECK(mf.seqs["Q"].size() == n * n && mf.seqs["P"].size() == n * n && mf.seqs["R"].size() == m * m);
    CHECK(mf.seqs["R"][0] == 1e-4 && mf.seqs["P"][0] == 0.1);
    std::printf("model %-22s n %2zu m %zu Q00 %.3e\n", mf.type.c_str(), n, m, mf.seqs["Q"][0]);
  }
  ModelFile bad;
  std::string err;
  CHECK(!load_model_file("/nonexistent/file.yaml", bad, err) && !err.empty());
  std::printf("%s\n", failures ? "HOST TESTS FAILED" : "host tests ok");
  return failures ? 1 : 0;
}
