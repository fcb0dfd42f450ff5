// The device-memory arena of the library (round 6): one per device, shared by every context of the process.
//
// Why: in the whole-recursion workload (BASELINE.json configs[4], 100 000 taxa) the driver's allocator was on the
// critical path -- hipMalloc of memory the process had used before costs ~24 ms per GB there (80 GB for the
// root's W: 1.9 s), single calls stalled for 2 s while a look-ahead worker's kernels ran, and every hipFree is
// a device-wide synchronisation (1.9 s behind a 2 s kernel: tools/probes/alloc_stall_probe.hip).  Together 7.6 s
// of a 45 s recursion on the main thread alone (profiles/r06_alloc_trace.txt).  So memory taken from the driver
// stays here as SLABS, requests are CARVED out of free slab space (best fit, address-ordered coalescing) and go
// back to it, and the driver is only called when no free chunk is large enough -- or to hand whole free slabs
// back (scs_ctx_trim, or a failed driver allocation).
//
// Stream safety: a released chunk is PENDING at first -- its owner context may take it again at once (its own
// stream order protects the new use, as the per-context block cache did before; no HIP call on that path), any
// other context only once the owner's streams have passed the point of the release.  That is established per
// OWNER, not per chunk, and only when somebody needs it (`poll`): the owner's streams are idle (one query), or a
// marker -- events recorded on them later than the release -- has completed.
//
// This header is the allocator's LOGIC only (no HIP types): the backing allocation and the stream state are
// callbacks, so tests/test_arena_cpu.py drives the same code with malloc and scripted streams on a machine
// without a GPU.
#pragma once
#include <cstddef>
#include <cstdint>
#include <deque>
#include <functional>
#include <map>
#include <utility>
#include <vector>

struct scs_arena_core {
    static constexpr size_t SMALL_LIMIT = (size_t)1 << 20;       // requests below this live in slabs of their own
    static constexpr size_t SMALL_SLAB = (size_t)32 << 20;
    static constexpr size_t LARGE_SLAB_MIN = (size_t)256 << 20;
    static constexpr size_t SMALL_GRAIN = 512, LARGE_GRAIN = 4096;
    static constexpr size_t LARGE_MIN_SPLIT = (size_t)64 << 10;  // a smaller remainder stays with the chunk
    static constexpr size_t POLL_EVERY = 512;                    // pending chunks before a release polls

    struct chunk {
        size_t bytes = 0;
        int slab = -1;
        bool free = false;   // released by its user
        bool ready = false;  // ... and safe for every context (in ready_by_size)
        const void *owner = nullptr;
        uint64_t seq = 0;  // order of the release (pending chunks)
    };
    struct slab_t {
        char *base = nullptr;
        size_t bytes = 0;
        bool small = false;
    };
    struct marker {
        uint64_t seq;
        std::vector<void *> events;
    };
    struct owner_state {
        std::multimap<size_t, char *> pend[2];  // its pending chunks by size: [0] large slabs, [1] small slabs
        std::deque<marker> markers;
        uint64_t safe_seq = 0;  // releases up to here are complete
    };

    // ---- backing (set by the owner of the arena)
    std::function<void *(size_t)> back_alloc;  // nullptr when the driver has no room
    std::function<void(void *)> back_free;
    // events on the owner's streams that still have work queued (none: the owner is idle)
    std::function<std::vector<void *>(const void *owner)> owner_mark;
    std::function<bool(void *)> event_done;
    std::function<void(void *)> event_wait;
    std::function<void(void *)> event_recycle;

    std::map<char *, chunk> chunks;  // every chunk of every slab, by address
    std::multimap<size_t, char *> ready_by_size[2];
    std::map<const void *, owner_state> owners;
    std::vector<slab_t> slabs;
    size_t slab_bytes = 0, used_bytes = 0, n_pending = 0;
    uint64_t seq = 0;
    uint64_t n_driver_allocs = 0, n_driver_frees = 0, n_allocs = 0, n_pending_reuse = 0, n_polls = 0;

    static size_t round_up(size_t bytes) {
        if (bytes == 0) bytes = 1;
        const size_t g = bytes < SMALL_LIMIT ? SMALL_GRAIN : LARGE_GRAIN;
        return (bytes + g - 1) / g * g;
    }

    size_t free_bytes() const { return slab_bytes - used_bytes; }

    // ------------------------------------------------------------------ internals
    using chunk_it = std::map<char *, chunk>::iterator;

    bool is_small(const chunk &c) const { return slabs[c.slab].small; }

    static void erase_from(std::multimap<size_t, char *> &idx, size_t bytes, char *p) {
        auto range = idx.equal_range(bytes);
        for (auto it = range.first; it != range.second; ++it)
            if (it->second == p) {
                idx.erase(it);
                return;
            }
    }

    // take a free chunk out of whichever index holds it
    void unindex(chunk_it it) {
        chunk &c = it->second;
        if (c.ready) {
            erase_from(ready_by_size[is_small(c) ? 1 : 0], c.bytes, it->first);
        } else {
            erase_from(owners[c.owner].pend[is_small(c) ? 1 : 0], c.bytes, it->first);
            --n_pending;
        }
    }

    void index(chunk_it it) {
        chunk &c = it->second;
        if (c.ready) {
            ready_by_size[is_small(c) ? 1 : 0].emplace(c.bytes, it->first);
        } else {
            owners[c.owner].pend[is_small(c) ? 1 : 0].emplace(c.bytes, it->first);
            ++n_pending;
        }
    }

    // two free neighbours merge when both are ready, or both pending of the same owner (the later release counts)
    static bool mergeable(const chunk &a, const chunk &b) {
        if (!a.free || !b.free || a.slab != b.slab || a.ready != b.ready) return false;
        return a.ready || a.owner == b.owner;
    }

    // a free chunk whose state is final: merge with its neighbours, index the result
    void settle(chunk_it it) {
        auto next = std::next(it);
        if (next != chunks.end() && it->first + it->second.bytes == next->first && mergeable(it->second, next->second)) {
            unindex(next);
            it->second.bytes += next->second.bytes;
            if (next->second.seq > it->second.seq) it->second.seq = next->second.seq;
            chunks.erase(next);
        }
        if (it != chunks.begin()) {
            auto prev = std::prev(it);
            if (prev->first + prev->second.bytes == it->first && mergeable(prev->second, it->second)) {
                unindex(prev);
                prev->second.bytes += it->second.bytes;
                if (it->second.seq > prev->second.seq) prev->second.seq = it->second.seq;
                chunks.erase(it);
                it = prev;
            }
        }
        index(it);
    }

    // take `need` bytes from the front of a free chunk (ready, or pending of `owner`)
    void *carve(chunk_it it, size_t need, const void *owner) {
        unindex(it);
        chunk &c = it->second;
        const bool small = is_small(c);
        const size_t min_split = small ? SMALL_GRAIN : LARGE_MIN_SPLIT;
        if (c.bytes >= need + min_split) {
            chunk rest = c;  // (same state: ready, or pending of the same owner with the same release order)
            rest.bytes = c.bytes - need;
            c.bytes = need;
            index(chunks.emplace(it->first + need, rest).first);
        }
        c.free = c.ready = false;
        c.owner = owner;
        used_bytes += c.bytes;
        return it->first;
    }

    void *from_free(size_t need, bool small, const void *owner) {
        // the smaller of: the best ready chunk, the best pending chunk of this owner
        auto &ridx = ready_by_size[small ? 1 : 0];
        auto r = ridx.lower_bound(need);
        auto os = owners.find(owner);
        if (os != owners.end()) {
            auto &pidx = os->second.pend[small ? 1 : 0];
            auto p = pidx.lower_bound(need);
            if (p != pidx.end() && (r == ridx.end() || p->first <= r->first)) {
                ++n_pending_reuse;
                return carve(chunks.find(p->second), need, owner);
            }
        }
        if (r == ridx.end()) return nullptr;
        return carve(chunks.find(r->second), need, owner);
    }

    // which releases are complete by now (`wait`: block until all of them are); promote their chunks
    void poll(bool wait) {
        ++n_polls;
        for (auto &kv : owners) {
            owner_state &st = kv.second;
            if (st.pend[0].empty() && st.pend[1].empty() && st.markers.empty()) continue;
            while (!st.markers.empty()) {
                marker &m = st.markers.front();
                bool done = true;
                for (void *e : m.events) {
                    if (wait) event_wait(e);
                    else if (!event_done(e)) {
                        done = false;
                        break;
                    }
                }
                if (!done) break;
                for (void *e : m.events) event_recycle(e);
                if (m.seq > st.safe_seq) st.safe_seq = m.seq;
                st.markers.pop_front();
            }
            uint64_t newest = 0;
            for (auto &idx : st.pend)
                for (auto &e : idx) {
                    const uint64_t s = chunks.find(e.second)->second.seq;
                    if (s > newest) newest = s;
                }
            if (newest > st.safe_seq && (st.markers.empty() || st.markers.back().seq < newest)) {
                std::vector<void *> ev = owner_mark(kv.first);
                if (ev.empty()) {
                    st.safe_seq = seq;  // its streams are idle: everything released so far is complete
                } else if (wait) {
                    for (void *e : ev) {
                        event_wait(e);
                        event_recycle(e);
                    }
                    st.safe_seq = seq;
                } else {
                    st.markers.push_back({seq, std::move(ev)});
                }
            }
            for (auto &idx : st.pend) {
                std::vector<char *> up;
                for (auto &e : idx)
                    if (chunks.find(e.second)->second.seq <= st.safe_seq) up.push_back(e.second);
                for (char *p : up) {
                    auto it = chunks.find(p);
                    if (it == chunks.end() || it->second.ready) continue;  // (merged into a neighbour just promoted)
                    unindex(it);
                    it->second.ready = true;
                    it->second.owner = nullptr;
                    settle(it);
                }
            }
        }
    }

    // hand whole free slabs back to the driver, largest first, until at most `keep` free bytes remain
    size_t release_slabs(size_t keep) {
        size_t released = 0;
        while (free_bytes() > keep) {
            int pick = -1;
            for (size_t s = 0; s < slabs.size(); ++s) {
                if (!slabs[s].base) continue;
                auto it = chunks.find(slabs[s].base);
                if (it != chunks.end() && it->second.ready && it->second.bytes == slabs[s].bytes &&
                    (pick < 0 || slabs[s].bytes > slabs[pick].bytes))
                    pick = (int)s;
            }
            if (pick < 0) break;
            auto it = chunks.find(slabs[pick].base);
            unindex(it);
            chunks.erase(it);
            back_free(slabs[pick].base);
            ++n_driver_frees;
            slab_bytes -= slabs[pick].bytes;
            released += slabs[pick].bytes;
            slabs[pick] = slab_t();
        }
        return released;
    }

    // ------------------------------------------------------------------ interface (the caller holds the arena's lock)
    void *alloc(size_t bytes, const void *owner) {
        const size_t need = round_up(bytes);
        const bool small = need < SMALL_LIMIT;
        ++n_allocs;
        if (void *p = from_free(need, small, owner)) return p;
        if (n_pending) {
            poll(false);
            if (void *p = from_free(need, small, owner)) return p;
        }
        // a new slab
        size_t want = small ? (need > SMALL_SLAB ? need : SMALL_SLAB) : (need > LARGE_SLAB_MIN ? need : LARGE_SLAB_MIN);
        char *base = (char *)back_alloc(want);
        if (!base && want > need) {
            want = need;
            base = (char *)back_alloc(want);
        }
        if (!base) {
            // make room: everything released anywhere becomes ready, whole free slabs go back, once more
            poll(true);
            if (void *p = from_free(need, small, owner)) return p;
            release_slabs(0);
            base = (char *)back_alloc(want);
            if (!base) return nullptr;
        }
        ++n_driver_allocs;
        int s = -1;
        for (size_t i = 0; i < slabs.size(); ++i)
            if (!slabs[i].base) {
                s = (int)i;
                break;
            }
        if (s < 0) {
            slabs.emplace_back();
            s = (int)slabs.size() - 1;
        }
        slabs[s].base = base;
        slabs[s].bytes = want;
        slabs[s].small = small;
        slab_bytes += want;
        chunk c;
        c.bytes = want;
        c.slab = s;
        c.free = c.ready = true;
        auto it = chunks.emplace(base, c).first;
        index(it);
        return carve(it, need, owner);
    }

    bool holds(void *p) const {
        auto it = chunks.find((char *)p);
        return it != chunks.end() && !it->second.free;
    }

    bool release(void *p) {
        auto it = chunks.find((char *)p);
        if (it == chunks.end() || it->second.free) return false;
        used_bytes -= it->second.bytes;
        it->second.free = true;
        if (it->second.owner == nullptr) {
            it->second.ready = true;
        } else {
            it->second.ready = false;
            it->second.seq = ++seq;
        }
        settle(it);
        if (n_pending > POLL_EVERY) poll(false);
        return true;
    }

    // a context goes away (its streams are idle): its pending chunks become ready; what objects made on it still
    // hold (a forest, a graph freed after its context) is ORPHANED, not released -- a later release of such a chunk
    // is then the first and only one, instead of one that hits whoever was given the address in between
    void owner_gone(const void *owner) {
        for (auto &kv : chunks)
            if (!kv.second.free && kv.second.owner == owner) kv.second.owner = nullptr;
        auto os = owners.find(owner);
        if (os == owners.end()) return;
        for (auto &m : os->second.markers)
            for (void *e : m.events) event_recycle(e);
        os->second.markers.clear();
        for (auto &idx : os->second.pend) {
            std::vector<char *> up;
            for (auto &e : idx) up.push_back(e.second);
            for (char *p : up) {
                auto it = chunks.find(p);
                if (it == chunks.end() || it->second.ready) continue;
                unindex(it);
                it->second.ready = true;
                it->second.owner = nullptr;
                settle(it);
            }
        }
        owners.erase(owner);
    }

    size_t trim(size_t keep) {
        if (n_pending) poll(false);
        return release_slabs(keep);
    }

    // consistency of the maps (tests): every byte of every slab in exactly one chunk, the indices complete
    bool check() const {
        size_t used = 0, total = 0, pend = 0;
        const chunk *prev = nullptr;
        const char *prev_p = nullptr;
        for (auto &kv : chunks) {
            const chunk &c = kv.second;
            if (c.slab < 0 || (size_t)c.slab >= slabs.size() || !slabs[c.slab].base) return false;
            const slab_t &s = slabs[c.slab];
            if (kv.first < s.base || kv.first + c.bytes > s.base + s.bytes) return false;
            if (prev && prev->slab == c.slab && prev_p + prev->bytes != kv.first) return false;
            if ((!prev || prev->slab != c.slab) && kv.first != s.base) return false;
            if (prev && prev->slab == c.slab && mergeable(*prev, c)) return false;  // (should have been merged)
            if (!c.free) used += c.bytes;
            if (c.free && !c.ready) ++pend;
            total += c.bytes;
            prev = &c;
            prev_p = kv.first;
        }
        size_t in_ready = ready_by_size[0].size() + ready_by_size[1].size(), n_ready = 0, in_pend = 0;
        for (auto &kv : chunks) n_ready += kv.second.free && kv.second.ready;
        for (auto &o : owners) in_pend += o.second.pend[0].size() + o.second.pend[1].size();
        return used == used_bytes && total == slab_bytes && pend == n_pending && in_ready == n_ready && in_pend == pend;
    }
};
