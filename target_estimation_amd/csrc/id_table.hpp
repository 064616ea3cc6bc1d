// id_table.hpp -- target id -> (batch, slot) of the TargetManager mirror.
//
// The reference keeps a std::map<unsigned, TargetPtr> (target_manager.hpp:201) and looks an id up on
// every call; a by-id batch call of 10^6 targets would spend a third of a second in red-black-tree
// lookups.  Here: an open-addressing table (linear probing, power-of-two size, load <= 1/2,
// backward-shift deletion), about one cache line per lookup.  The reference's ascending enumeration
// order (getAvailableTargets, target_manager.cpp:126-133) is produced on demand by sorting.
#pragma once
#include <algorithm>
#include <cstddef>
#include <vector>

namespace te {

struct TargetLoc { int batch; int slot; };

class IdTable {
 public:
  IdTable() { rebuild(16); }
  size_t size() const { return n_; }
  bool contains(unsigned id) const { return tab_[probe(id)].batch >= 0; }
  bool find(unsigned id, TargetLoc& out) const {
    const Entry& e = tab_[probe(id)];
    if (e.batch < 0) return false;
    out.batch = e.batch; out.slot = e.slot;
    return true;
  }
  // insert or overwrite
  void set(unsigned id, TargetLoc loc) {
    if ((n_ + 1) * 2 > tab_.size()) rebuild(tab_.size() * 2);
    Entry& e = tab_[probe(id)];
    if (e.batch < 0) ++n_;
    e.id = id; e.batch = loc.batch; e.slot = loc.slot;
  }
  void reserve(size_t n) {
    size_t cap = tab_.size();
    while (n * 2 > cap) cap *= 2;
    if (cap != tab_.size()) rebuild(cap);
  }
  bool erase(unsigned id) {
    size_t i = probe(id);
    if (tab_[i].batch < 0) return false;
    // backward-shift deletion: close the gap so that probe chains stay unbroken
    const size_t mask = tab_.size() - 1;
    size_t j = i;
    for (;;) {
      j = (j + 1) & mask;
      if (tab_[j].batch < 0) break;
      const size_t home = hash(tab_[j].id);
      // entry j may move into the hole i if its home position is cyclically outside (i, j]
      const bool between = (i <= j) ? (home > i && home <= j) : (home > i || home <= j);
      if (!between) { tab_[i] = tab_[j]; i = j; }
    }
    tab_[i].batch = -1;
    --n_;
    return true;
  }
  std::vector<unsigned> sorted_ids() const {
    std::vector<unsigned> ids;
    ids.reserve(n_);
    for (const Entry& e : tab_) if (e.batch >= 0) ids.push_back(e.id);
    std::sort(ids.begin(), ids.end());
    return ids;
  }

 private:
  struct Entry { unsigned id; int batch; int slot; };   // batch < 0: empty
  size_t hash(unsigned id) const { return (size_t)((id * 2654435761u) >> shift_); }
  // position of id, or of the empty slot where it would go
  size_t probe(unsigned id) const {
    const size_t mask = tab_.size() - 1;
    size_t i = hash(id);
    while (tab_[i].batch >= 0 && tab_[i].id != id) i = (i + 1) & mask;
    return i;
  }
  void rebuild(size_t cap) {
    std::vector<Entry> old;
    old.swap(tab_);
    tab_.assign(cap, Entry{0u, -1, 0});
    unsigned bits = 0;
    while (((size_t)1 << bits) < cap) ++bits;
    shift_ = 32 - bits;
    n_ = 0;
    for (const Entry& e : old)
      if (e.batch >= 0) { tab_[probe(e.id)] = e; ++n_; }
  }
  std::vector<Entry> tab_;
  size_t n_ = 0;
  unsigned shift_ = 28;
};

}  // namespace te
