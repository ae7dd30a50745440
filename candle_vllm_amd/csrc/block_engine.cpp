// Host-side KV block manager: the producer of the block tables / slot mappings the kernels consume.
// SURVEY.md section 8(f1).  Behavioural contract restated from the reference (not its data structures):
//   BlockEngine      src/scheduler/block_engine.rs:190-1474   (FIFO allocators, ref-counted blocks, COW on append,
//                    chunked-prefill reservation, swap in/out with rollback, prefix-cache reuse)
//   PrefixCache      src/scheduler/prefix_cache.rs:36-384     (hash-chained full blocks, LRU-leaf eviction,
//                    just-inserted blocks protected)
//   Sequence         src/scheduler/sequence.rs:90-300         (logical blocks = len/bs + 1, prefill_chunk_tokens)
//   prepare_decode / prepare_prompt   src/openai/pipelines/inputs.rs:90-230,376-454
// Observable behaviour = block ids, free counts, cached-token counts, eviction order -- pinned by the reference's own
// unit tests (block_engine.rs:1476-1752, prefix_cache.rs:386-599, sequence.rs:479-537), re-stated in
// tests/test_cpu_block_engine.py.  The chain hash is Rust's DefaultHasher in the reference and is not observable;
// a 64-bit mix is used here.  Mamba snapshot bookkeeping and image seeds belong to models outside the hot path.
//
// Representation: a physical block is an int32 code: id >= 0 on the GPU, -1-id on the CPU.  Reference counts live
// in two flat arrays; free lists are FIFO deques (allocate = pop_front, free = push_back), which is what fixes the
// ids a fresh allocation returns.
#include <cstdint>
#include <cstring>
#include <deque>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include <algorithm>

#include "../../include/mi355_vllm.h"

namespace {

typedef int32_t Blk;
inline bool is_gpu(Blk b) { return b >= 0; }
inline int cpu_id(Blk b) { return -1 - b; }
inline Blk cpu_code(int id) { return -1 - id; }

struct Pool {
    std::deque<int> free_ids;
    std::vector<uint32_t> ref;
    void init(int n) { ref.assign(n, 0); free_ids.clear(); for (int i = 0; i < n; ++i) free_ids.push_back(i); }
    int allocate() { const int id = free_ids.front(); free_ids.pop_front(); ref[id] = 1; return id; }
    bool release(int id) {                                   // false = double free
        if (ref[id] == 0) return false;
        if (--ref[id] == 0) free_ids.push_back(id);
        return true;
    }
    void retain(int id) {                                    // block_engine.rs:904-922
        if (ref[id]++ == 0) free_ids.erase(std::remove(free_ids.begin(), free_ids.end(), id), free_ids.end());
    }
};

inline uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
    return x;
}
uint64_t hash_block(uint64_t parent, const uint32_t* tok, int n) {
    uint64_t h = mix64(parent ^ 0x9e3779b97f4a7c15ull);
    for (int i = 0; i < n; ++i) h = mix64(h ^ (uint64_t)tok[i] ^ ((uint64_t)(i + 1) << 32));
    return h ? h : 1;                                        // 0 is the "no parent" value
}
uint64_t mix_seed(uint64_t parent, uint64_t seed) { return mix64(mix64(parent) ^ seed ^ 0xd6e8feb86659fd93ull); }

struct PrefixEntry { bool has_parent; uint64_t parent; int block; size_t children; uint64_t access_id; };

struct PrefixCache {
    int block_size = 0; bool cfg_enabled = false; size_t max_blocks = 0;
    std::unordered_map<uint64_t, PrefixEntry> entries;
    std::unordered_set<uint64_t> leaf_set;
    std::deque<std::pair<uint64_t, uint64_t>> leaf_lru;
    uint64_t access_counter = 0;

    bool enabled() const { return cfg_enabled && max_blocks > 0; }
    uint64_t next_access() { return ++access_counter; }
    void compact() {                                         // prefix_cache.rs:307-321
        const size_t threshold = std::max<size_t>(entries.size(), 64) * 4;
        if (leaf_lru.size() <= threshold) return;
        std::deque<std::pair<uint64_t, uint64_t>> keep;
        for (auto& it : leaf_lru) {
            if (!leaf_set.count(it.first)) continue;
            auto e = entries.find(it.first);
            if (e != entries.end() && e->second.access_id == it.second) keep.push_back(it);
        }
        leaf_lru.swap(keep);
    }
    void touch_leaf(uint64_t h) {
        if (leaf_set.count(h)) { auto e = entries.find(h); if (e != entries.end()) leaf_lru.push_back({h, e->second.access_id}); }
        compact();
    }
    void touch(uint64_t h) {
        auto e = entries.find(h);
        if (e == entries.end()) return;
        e->second.access_id = next_access();
        touch_leaf(h);
    }
    // -> matched full blocks; last_hash valid when > 0
    int match(const uint32_t* tok, int n, bool has_seed, uint64_t seed, int seed_block, uint64_t* last_hash) {
        if (!enabled()) return 0;
        const int full = n / block_size;
        uint64_t parent = 0; int matched = 0;
        for (int i = 0; i < full; ++i) {
            if (has_seed && i == seed_block) parent = mix_seed(parent, seed);
            const uint64_t h = hash_block(parent, tok + (size_t)i * block_size, block_size);
            if (!entries.count(h)) break;
            ++matched; parent = h; *last_hash = h; touch(h);
        }
        return matched;
    }
    bool hash_for_blocks(const uint32_t* tok, int n, int full_blocks, bool has_seed, uint64_t seed, int seed_block, uint64_t* out) const {
        if (!enabled() || full_blocks == 0) return false;
        uint64_t parent = 0; bool any = false;
        const int avail = (n + block_size - 1) / block_size;
        for (int i = 0; i < full_blocks && i < avail; ++i) {
            if (has_seed && i == seed_block) parent = mix_seed(parent, seed);
            const int len = std::min(block_size, n - i * block_size);
            parent = hash_block(parent, tok + (size_t)i * block_size, len);
            *out = parent; any = true;
        }
        return any;
    }
    std::vector<int> blocks_for_match(uint64_t last_hash) const {
        std::vector<int> b;
        bool has = true; uint64_t cur = last_hash;
        while (has) {
            auto e = entries.find(cur);
            if (e == entries.end()) break;
            b.push_back(e->second.block);
            has = e->second.has_parent; cur = e->second.parent;
        }
        std::reverse(b.begin(), b.end());
        return b;
    }
    bool evict_leaf_hash(uint64_t h, int* block) {
        auto e = entries.find(h);
        if (e == entries.end()) return false;
        const PrefixEntry ent = e->second;
        entries.erase(e);
        leaf_set.erase(h);
        if (ent.has_parent) {
            auto p = entries.find(ent.parent);
            if (p != entries.end()) {
                if (p->second.children > 0) --p->second.children;
                if (p->second.children == 0) { leaf_set.insert(ent.parent); leaf_lru.push_back({ent.parent, p->second.access_id}); }
            }
        }
        *block = ent.block;
        return true;
    }
    bool evict_one(const std::unordered_set<uint64_t>& prot, int* block) {      // prefix_cache.rs:323-352
        std::deque<std::pair<uint64_t, uint64_t>> skipped;
        bool ok = false;
        while (!leaf_lru.empty()) {
            const auto it = leaf_lru.front(); leaf_lru.pop_front();
            if (!leaf_set.count(it.first)) continue;
            auto e = entries.find(it.first);
            if (e == entries.end()) continue;
            if (e->second.access_id != it.second || e->second.children > 0) continue;
            if (prot.count(it.first)) { skipped.push_back(it); continue; }
            ok = evict_leaf_hash(it.first, block);
            break;
        }
        for (auto r = skipped.rbegin(); r != skipped.rend(); ++r) leaf_lru.push_front(*r);
        return ok;
    }
    std::vector<int> evict(size_t n, const std::unordered_set<uint64_t>& prot) {
        std::vector<int> out;
        while (out.size() < n) { int b; if (!evict_one(prot, &b)) break; out.push_back(b); }
        return out;
    }
    // insert the full blocks of `tok` backed by `blocks` (GPU ids); refs[] is bumped for newly cached blocks.
    std::vector<int> insert(const uint32_t* tok, int n, const int* blocks, int nblocks, std::vector<uint32_t>& refs,
                            bool has_seed, uint64_t seed, int seed_block) {
        std::vector<int> evicted;
        if (!enabled()) return evicted;
        const int maxb = std::min(n / block_size, nblocks);
        if (maxb == 0) return evicted;
        bool has_parent = false; uint64_t parent = 0;
        std::unordered_set<uint64_t> prot;
        for (int i = 0; i < maxb; ++i) {
            uint64_t base = has_parent ? parent : 0;
            if (has_seed && i == seed_block) base = mix_seed(base, seed);
            const uint64_t h = hash_block(base, tok + (size_t)i * block_size, block_size);
            prot.insert(h);
            auto e = entries.find(h);
            if (e != entries.end()) {
                e->second.access_id = next_access();
                touch_leaf(h);
            } else {
                if (has_parent) {
                    auto p = entries.find(parent);
                    if (p != entries.end()) { if (p->second.children == 0) leaf_set.erase(parent); ++p->second.children; }
                }
                ++refs[blocks[i]];
                const uint64_t a = next_access();
                entries[h] = PrefixEntry{has_parent, parent, blocks[i], 0, a};
                leaf_set.insert(h);
                leaf_lru.push_back({h, a});
            }
            has_parent = true; parent = h;
        }
        if (entries.size() > max_blocks) evicted = evict(entries.size() - max_blocks, prot);
        return evicted;
    }
};

struct Seq {
    std::vector<uint32_t> tokens;
    int prompt_len = 0, num_cached = 0;
    bool has_warmup = false; int warmup = 0;
    bool has_prefix_hash = false; uint64_t prefix_hash = 0;
};

struct Pending { std::vector<std::pair<int64_t, std::vector<Blk>>> old_tables, new_tables; };

struct Engine {
    int block_size = 0, num_gpu = 0;
    Pool gpu, cpu;
    bool has_cache = false;
    PrefixCache cache;
    std::unordered_map<int64_t, std::vector<Blk>> tables;
    std::unordered_map<int64_t, Seq> seqs;
    std::unordered_map<int64_t, Pending> pending_out, pending_in;
    int refuse_swap_out = 0, refuse_swap_in = 0;          // test hook (mi355_be_test_refuse_swaps): the next n swaps are refused up front

    int logical_blocks(const Seq& s) const { return (int)s.tokens.size() / block_size + 1; }     // sequence.rs:208-227
    int blocks_to_add_new_tok(const Seq& s) const { return (s.tokens.size() % block_size) == 0 ? 1 : 0; }
    int prefill_chunk_tokens(const Seq& s, int chunk) const {                                       // sequence.rs:279-299
        const int remaining = std::max(0, s.prompt_len - s.num_cached);
        if (remaining == 0) return 0;
        if (chunk == 0) return remaining;
        if (s.has_warmup && s.warmup > s.num_cached && s.warmup < s.prompt_len) return std::min(s.warmup - s.num_cached, chunk);
        return std::min(remaining, chunk);
    }
    static int chunk_end(int prompt_len, int cached, int chunk) {                                   // block_engine.rs:417-439
        cached = std::min(cached, prompt_len);
        if (chunk == 0) return prompt_len;
        return cached + std::min(std::max(0, prompt_len - cached), chunk);
    }
    void release(Blk b) { if (is_gpu(b)) gpu.release(b); else cpu.release(cpu_id(b)); }
    void retain(Blk b) { if (is_gpu(b)) gpu.retain(b); else cpu.retain(cpu_id(b)); }
    uint32_t refcount(Blk b) const { return is_gpu(b) ? gpu.ref[b] : cpu.ref[cpu_id(b)]; }

    int evict_until_free(int min_free) {                                                            // :648-674
        int total = 0;
        while ((int)gpu.free_ids.size() < min_free && has_cache) {
            auto ev = cache.evict(1, {});
            if (ev.empty()) break;
            total += (int)ev.size();
            for (int b : ev) gpu.release(b);
        }
        return total;
    }
    // prefix match for the first sequence of a group, with the "never reuse the whole prompt" rule (:312-320)
    int matched_blocks_for(const Seq& s, uint64_t* last_hash) {
        uint64_t lh = 0;
        const int n = (int)s.tokens.size();
        const int matched = cache.match(s.tokens.data(), n, false, 0, 0, &lh);
        const int full = n / block_size;
        *last_hash = lh;
        return (matched == full && n % block_size == 0 && matched > 0) ? matched - 1 : matched;
    }
};

Engine* E(void* p) { return static_cast<Engine*>(p); }
Seq* find_seq(Engine* e, int64_t id) { auto it = e->seqs.find(id); return it == e->seqs.end() ? nullptr : &it->second; }

}  // namespace

extern "C" {

void* mi355_be_create(int32_t block_size, int32_t num_gpu_blocks, int32_t num_cpu_blocks, int32_t prefix_cache_enabled,
                      int32_t max_cached_blocks) {
    if (block_size <= 0 || num_gpu_blocks < 0 || num_cpu_blocks < 0) return nullptr;
    Engine* e = new Engine();
    e->block_size = block_size; e->num_gpu = num_gpu_blocks;
    e->gpu.init(num_gpu_blocks); e->cpu.init(num_cpu_blocks);
    e->has_cache = prefix_cache_enabled && max_cached_blocks > 0;                                   // :250-254
    e->cache.block_size = block_size; e->cache.cfg_enabled = prefix_cache_enabled != 0;
    e->cache.max_blocks = (size_t)std::max(0, max_cached_blocks);
    return e;
}
void mi355_be_destroy(void* be) { delete E(be); }

int32_t mi355_be_num_free_blocks(void* be) { return (int32_t)E(be)->gpu.free_ids.size(); }
int32_t mi355_be_num_free_cpu_blocks(void* be) { return (int32_t)E(be)->cpu.free_ids.size(); }
int32_t mi355_be_num_blocks(void* be) { return E(be)->num_gpu; }
int32_t mi355_be_prefix_cache_blocks(void* be) { return E(be)->has_cache ? (int32_t)E(be)->cache.entries.size() : 0; }
/* free GPU block ids in allocation order */
int32_t mi355_be_free_block_ids(void* be, int32_t* out, int32_t cap) {
    Engine* e = E(be); int n = 0;
    for (int id : e->gpu.free_ids) { if (n < cap) out[n] = id; ++n; }
    return n;
}

/* ---- sequences (sequence.rs:90-300) */
int32_t mi355_be_seq_create(void* be, int64_t seq_id, const uint32_t* prompt, int32_t n) {
    Engine* e = E(be);
    if (n < 0 || e->seqs.count(seq_id)) return -1;
    Seq s; s.tokens.assign(prompt, prompt + n); s.prompt_len = n;
    e->seqs[seq_id] = std::move(s);
    return 0;
}
int32_t mi355_be_seq_remove(void* be, int64_t seq_id) { return E(be)->seqs.erase(seq_id) ? 0 : -1; }
int32_t mi355_be_seq_add_token(void* be, int64_t seq_id, uint32_t token) {
    Seq* s = find_seq(E(be), seq_id); if (!s) return -1;
    s->tokens.push_back(token); return 0;
}
int32_t mi355_be_seq_len(void* be, int64_t seq_id) { Seq* s = find_seq(E(be), seq_id); return s ? (int32_t)s->tokens.size() : -1; }
int32_t mi355_be_seq_logical_blocks(void* be, int64_t seq_id) { Seq* s = find_seq(E(be), seq_id); return s ? E(be)->logical_blocks(*s) : -1; }
int32_t mi355_be_seq_get_cached_tokens(void* be, int64_t seq_id) { Seq* s = find_seq(E(be), seq_id); return s ? s->num_cached : -1; }
int32_t mi355_be_seq_set_cached_tokens(void* be, int64_t seq_id, int32_t n) { Seq* s = find_seq(E(be), seq_id); if (!s) return -1; s->num_cached = n; return 0; }
int32_t mi355_be_seq_set_warmup_tokens(void* be, int64_t seq_id, int32_t n) {     /* n < 0 clears */
    Seq* s = find_seq(E(be), seq_id); if (!s) return -1;
    s->has_warmup = n >= 0; s->warmup = n; return 0;
}
int32_t mi355_be_seq_prefill_chunk_tokens(void* be, int64_t seq_id, int32_t chunk) {
    Seq* s = find_seq(E(be), seq_id); return s ? E(be)->prefill_chunk_tokens(*s, chunk) : -1;
}
int32_t mi355_be_seq_has_prefix_hash(void* be, int64_t seq_id) { Seq* s = find_seq(E(be), seq_id); return s ? (s->has_prefix_hash ? 1 : 0) : -1; }

/* ---- block tables */
int32_t mi355_be_block_table(void* be, int64_t seq_id, int32_t* out, int32_t cap) {
    Engine* e = E(be);
    auto it = e->tables.find(seq_id);
    if (it == e->tables.end()) return -1;
    const int n = (int)it->second.size();
    for (int i = 0; i < n && i < cap; ++i) out[i] = it->second[i];
    return n;
}
int32_t mi355_be_block_refcount(void* be, int32_t block_code) { return (int32_t)E(be)->refcount(block_code); }
/* test hook mirroring block_engine.rs:1657-1663: drop the last table entry and free it */
int32_t mi355_be_pop_back_block(void* be, int64_t seq_id) {
    Engine* e = E(be);
    auto it = e->tables.find(seq_id);
    if (it == e->tables.end() || it->second.empty()) return -1;
    const Blk b = it->second.back(); it->second.pop_back(); e->release(b);
    return 0;
}

/* can_allocate_for_prefill (:296-373): 0 = Ok, 1 = Later, 2 = Impossible */
int32_t mi355_be_can_allocate(void* be, const int64_t* seq_ids, int32_t n, int32_t chunk) {
    Engine* e = E(be);
    if (n < 1) return 2;
    int total_required = 0;
    for (int i = 0; i < n; ++i) { Seq* s = find_seq(e, seq_ids[i]); if (!s) return 2; total_required += e->logical_blocks(*s); }
    int required;
    if (e->has_cache) {
        Seq* s = find_seq(e, seq_ids[0]);
        uint64_t lh;
        const int matched = e->matched_blocks_for(*s, &lh);
        const int end = Engine::chunk_end((int)s->tokens.size(), matched * e->block_size, chunk);
        required = std::max(0, (end + e->block_size - 1) / e->block_size - matched);
    } else {
        required = 0;
        for (int i = 0; i < n; ++i) {
            Seq* s = find_seq(e, seq_ids[i]);
            const int end = Engine::chunk_end(s->prompt_len, 0, chunk);
            required += (end + e->block_size - 1) / e->block_size;
        }
    }
    if ((int)e->gpu.free_ids.size() < required) e->evict_until_free(required);
    if (e->num_gpu < total_required) return 2;
    return (int)e->gpu.free_ids.size() >= required ? 0 : 1;
}

/* allocate_for_prefill (:390-415) / allocate_with_prefix (:1331-1465) */
int32_t mi355_be_allocate(void* be, const int64_t* seq_ids, int32_t n, int32_t chunk) {
    Engine* e = E(be);
    if (n < 1) return -1;
    std::vector<Seq*> ss(n);
    for (int i = 0; i < n; ++i) { ss[i] = find_seq(e, seq_ids[i]); if (!ss[i]) return -1; }
    std::vector<Blk> table;
    int cached_tokens = 0;
    if (e->has_cache) {
        Seq* s = ss[0];
        uint64_t lh = 0;
        int matched = e->matched_blocks_for(*s, &lh);
        if (matched > 0) {
            std::vector<int> blocks = e->cache.blocks_for_match(lh);
            if ((int)blocks.size() < matched) matched = 0;
            else {
                blocks.resize(matched);
                // the hash of the last REUSED block (the full match minus the trimmed tail)
                uint64_t h = 0;
                e->cache.hash_for_blocks(s->tokens.data(), (int)s->tokens.size(), matched, false, 0, 0, &h);
                s->has_prefix_hash = true; s->prefix_hash = h;
                for (int b : blocks) { ++e->gpu.ref[b]; table.push_back(b); }
            }
        }
        if (matched == 0) s->has_prefix_hash = false;
        s->has_warmup = false;
        cached_tokens = matched * e->block_size;
        s->num_cached = cached_tokens;
        const int end = e->prefill_chunk_tokens(*s, chunk) + cached_tokens;
        const int required = chunk == 0 ? e->logical_blocks(*s) : (end + e->block_size - 1) / e->block_size;
        // `can_allocate` counted ceil(prefill_end / block_size) blocks (:348-349); an unchunked allocation takes the LOGICAL block
        // count, one more when the token count is a multiple of the block size (sequence.rs:208-227).  With exactly the counted
        // blocks free the reference unwraps an empty free list here (:113-117).  We first give the prefix cache's evictable
        // blocks back (the matched blocks are pinned by the references taken above), and refuse only if that is not enough.
        if (required - (int)table.size() > (int)e->gpu.free_ids.size()) e->evict_until_free(required - (int)table.size());
        if (required - (int)table.size() > (int)e->gpu.free_ids.size()) {
            for (Blk b : table) e->release(b);
            return -2;                                            /* the reference would panic on an empty free list */
        }
        while ((int)table.size() < required) table.push_back(e->gpu.allocate());
    } else {
        int blocks = 0;
        if (chunk == 0) { for (Seq* s : ss) blocks += e->logical_blocks(*s); }
        else for (Seq* s : ss) blocks += (Engine::chunk_end(s->prompt_len, 0, chunk) + e->block_size - 1) / e->block_size;
        if (blocks > (int)e->gpu.free_ids.size()) return -2;
        for (int i = 0; i < blocks; ++i) table.push_back(e->gpu.allocate());
    }
    for (int i = 0; i < n; ++i) {
        if (e->has_cache) { ss[i]->num_cached = cached_tokens; if (i > 0) { ss[i]->has_prefix_hash = ss[0]->has_prefix_hash; ss[i]->prefix_hash = ss[0]->prefix_hash; } }
        if (i > 0) for (Blk b : table) ++e->gpu.ref[b];           /* shared by every sequence of the group */
        e->tables[seq_ids[i]] = table;
    }
    return 0;
}

static int blocks_missing_for_sequence(Engine* e, int64_t id, const Seq& s) {                       // :487-496
    auto it = e->tables.find(id);
    const int have = it == e->tables.end() ? 0 : (int)it->second.size();
    return std::max(0, e->logical_blocks(s) - have);
}
static int blocks_required_to_append(Engine* e, int64_t id, const Seq& s) {                         // :498-516
    const int missing = blocks_missing_for_sequence(e, id, s);
    if (missing > 0) return missing;
    if (e->blocks_to_add_new_tok(s) > 0) return 0;
    auto it = e->tables.find(id);
    if (it == e->tables.end() || it->second.empty()) return 0;
    return e->refcount(it->second.back()) > 1 ? 1 : 0;
}
int32_t mi355_be_can_append_token(void* be, const int64_t* seq_ids, int32_t n) {                    // :477-485
    Engine* e = E(be); int req = 0;
    for (int i = 0; i < n; ++i) { Seq* s = find_seq(e, seq_ids[i]); if (!s) return -1; req += blocks_required_to_append(e, seq_ids[i], *s); }
    return req <= (int)e->gpu.free_ids.size() ? 1 : 0;
}
/* append_token_slot_to_seq (:1181-1212): 1 = copy-on-write happened (src, dst returned), 0 = no copy */
int32_t mi355_be_append_token_slot(void* be, int64_t seq_id, int32_t* cow_src, int32_t* cow_dst) {
    Engine* e = E(be);
    Seq* s = find_seq(e, seq_id);
    auto it = e->tables.find(seq_id);
    if (!s || it == e->tables.end()) return -1;
    const int missing = blocks_missing_for_sequence(e, seq_id, *s);
    if (missing > 0) {
        if (missing > (int)e->gpu.free_ids.size()) return -2;
        for (int i = 0; i < missing; ++i) it->second.push_back(e->gpu.allocate());
        return 0;
    }
    if (e->blocks_to_add_new_tok(*s) > 0) return 0;
    if (it->second.empty()) return -1;
    Blk& last = it->second.back();
    if (!is_gpu(last)) return -3;
    if (e->gpu.ref[last] == 1) return 0;
    if (e->gpu.free_ids.empty()) return -2;
    const int nb = e->gpu.allocate();
    e->gpu.release(last);
    if (cow_src) *cow_src = last;
    if (cow_dst) *cow_dst = nb;
    last = nb;
    return 1;
}

static int blocks_missing_for_prefill_chunk(Engine* e, int64_t id, const Seq& s, int chunk) {       // :559-574
    const int end = s.num_cached + e->prefill_chunk_tokens(s, chunk);
    auto it = e->tables.find(id);
    const int have = it == e->tables.end() ? 0 : (int)it->second.size();
    return std::max(0, (end + e->block_size - 1) / e->block_size - have);
}
int32_t mi355_be_prefill_chunk_blocks_required(void* be, const int64_t* seq_ids, int32_t n, int32_t chunk) {
    Engine* e = E(be); int req = 0;
    for (int i = 0; i < n; ++i) { Seq* s = find_seq(e, seq_ids[i]); if (!s) return -1; req += blocks_missing_for_prefill_chunk(e, seq_ids[i], *s, chunk); }
    return req;
}
int32_t mi355_be_can_append_prefill_chunk(void* be, const int64_t* seq_ids, int32_t n, int32_t chunk) {
    const int r = mi355_be_prefill_chunk_blocks_required(be, seq_ids, n, chunk);
    return r < 0 ? r : (r <= (int)E(be)->gpu.free_ids.size() ? 1 : 0);
}
int32_t mi355_be_append_prefill_chunk_slots(void* be, const int64_t* seq_ids, int32_t n, int32_t chunk) {
    Engine* e = E(be);
    for (int i = 0; i < n; ++i) {
        Seq* s = find_seq(e, seq_ids[i]); if (!s) return -1;
        const int add = blocks_missing_for_prefill_chunk(e, seq_ids[i], *s, chunk);
        if (add == 0) continue;
        auto it = e->tables.find(seq_ids[i]);
        if (it == e->tables.end()) return -1;
        if (add > (int)e->gpu.free_ids.size()) return -2;
        for (int k = 0; k < add; ++k) it->second.push_back(e->gpu.allocate());
    }
    return 0;
}

int32_t mi355_be_free_sequence(void* be, int64_t seq_id) {                                          // :576-592
    Engine* e = E(be);
    auto it = e->tables.find(seq_id);
    if (it == e->tables.end()) return -1;
    for (Blk b : it->second) e->release(b);
    e->tables.erase(it);
    return 0;
}
int32_t mi355_be_cache_sequence(void* be, int64_t seq_id) {                                         // :594-646
    Engine* e = E(be);
    if (!e->has_cache || !e->cache.enabled()) return 0;
    Seq* s = find_seq(e, seq_id); if (!s) return -1;
    const int full = (int)s->tokens.size() / e->block_size;
    if (full == 0) return 0;
    auto it = e->tables.find(seq_id);
    if (it == e->tables.end() || (int)it->second.size() < full) return 0;
    std::vector<int> blocks(it->second.begin(), it->second.begin() + full);
    for (int b : blocks) if (!is_gpu(b)) return 0;
    auto ev = e->cache.insert(s->tokens.data(), (int)s->tokens.size(), blocks.data(), full, e->gpu.ref, false, 0, 0);
    for (int b : ev) e->gpu.release(b);
    return (int32_t)ev.size();
}
int32_t mi355_be_evict_prefix_cache_blocks(void* be, int32_t num_blocks) {                          // :676-694
    Engine* e = E(be);
    if (!e->has_cache || num_blocks <= 0) return 0;
    auto ev = e->cache.evict((size_t)num_blocks, {});
    for (int b : ev) e->gpu.release(b);
    return (int32_t)ev.size();
}
int32_t mi355_be_evict_prefix_cache_until_free(void* be, int32_t min_free) { return E(be)->evict_until_free(min_free); }
int32_t mi355_be_query_prefix_match_tokens(void* be, const uint32_t* tokens, int32_t n) {            // :710-717
    Engine* e = E(be);
    if (!e->has_cache) return 0;
    uint64_t lh;
    return e->cache.match(tokens, n, false, 0, 0, &lh) * e->block_size;
}

/* rebuild a running sequence without (or with a shorter) shared prefix (:933-1038); 1 = rebuilt, 0 = not enough blocks */
int32_t mi355_be_fallback_to_full_prefill(void* be, int64_t seq_id) {
    Engine* e = E(be);
    Seq* s = find_seq(e, seq_id); if (!s) return -1;
    auto it = e->tables.find(seq_id);
    if (it == e->tables.end()) return 0;
    const int logical = e->logical_blocks(*s);
    std::vector<Blk> old = it->second;
    e->tables.erase(it);
    for (Blk b : old) e->release(b);
    if ((int)e->gpu.free_ids.size() < logical) e->evict_until_free(logical);
    if ((int)e->gpu.free_ids.size() < logical) {
        for (Blk b : old) e->retain(b);
        e->tables[seq_id] = old;
        return 0;
    }
    std::vector<Blk> nt;
    for (int i = 0; i < logical; ++i) nt.push_back(e->gpu.allocate());
    e->tables[seq_id] = nt;
    s->num_cached = 0; s->has_prefix_hash = false; s->has_warmup = false;
    return 1;
}
int32_t mi355_be_rebuild_with_cached_prefix(void* be, int64_t seq_id, int32_t cached_tokens) {
    Engine* e = E(be);
    Seq* s = find_seq(e, seq_id); if (!s) return -1;
    if (cached_tokens == 0) return mi355_be_fallback_to_full_prefill(be, seq_id);
    const int logical = e->logical_blocks(*s);
    const int full = cached_tokens / e->block_size;
    if (full == 0 || full > logical) return mi355_be_fallback_to_full_prefill(be, seq_id);
    uint64_t target = 0;
    if (!e->has_cache || !e->cache.hash_for_blocks(s->tokens.data(), (int)s->tokens.size(),
                                                    std::min(cached_tokens, (int)s->tokens.size()) / e->block_size, false, 0, 0, &target))
        return mi355_be_fallback_to_full_prefill(be, seq_id);
    auto it = e->tables.find(seq_id);
    if (it == e->tables.end()) return 0;
    std::vector<Blk> old = it->second;
    if ((int)old.size() < full) return 0;
    e->tables.erase(it);
    std::vector<Blk> prefix(old.begin(), old.begin() + full);
    for (Blk b : old) e->release(b);
    for (Blk b : prefix) e->retain(b);
    const int suffix = std::max(0, logical - full);
    if ((int)e->gpu.free_ids.size() < suffix) e->evict_until_free(suffix);
    if ((int)e->gpu.free_ids.size() < suffix) {
        for (Blk b : prefix) e->release(b);
        for (Blk b : old) e->retain(b);
        e->tables[seq_id] = old;
        return 0;
    }
    std::vector<Blk> nt = prefix;
    for (int i = full; i < logical; ++i) nt.push_back(e->gpu.allocate());
    e->tables[seq_id] = nt;
    s->num_cached = full * e->block_size;
    s->has_prefix_hash = true; s->prefix_hash = target; s->has_warmup = false;
    return 1;
}

/* ---- swap (:1040-1329).  Mappings come back as (src_id, dst_id) pairs, src/dst ids are plain block numbers. */
static std::unordered_map<int64_t, int> prefix_block_counts(Engine* e, const int64_t* ids, int n) {  // :1083-1106
    std::unordered_map<int64_t, int> counts;
    for (int i = 0; i < n; ++i) {
        auto it = e->tables.find(ids[i]);
        if (it == e->tables.end()) continue;
        Seq* s = find_seq(e, ids[i]);
        int cnt = e->has_cache && s ? s->num_cached / e->block_size : 0;
        cnt = std::min(cnt, (int)it->second.size());
        while (cnt < (int)it->second.size() && e->refcount(it->second[cnt]) > 1) ++cnt;
        counts[ids[i]] = cnt;
    }
    return counts;
}
int32_t mi355_be_can_swap_out(void* be, const int64_t* seq_ids, int32_t n) {                        // :1040-1077
    Engine* e = E(be);
    auto pc = prefix_block_counts(e, seq_ids, n);
    std::unordered_map<int, uint32_t> group_refs;
    for (int i = 0; i < n; ++i) {
        auto it = e->tables.find(seq_ids[i]);
        if (it == e->tables.end()) continue;
        for (size_t k = pc[seq_ids[i]]; k < it->second.size(); ++k) {
            if (!is_gpu(it->second[k])) return 0;
            ++group_refs[it->second[k]];
        }
    }
    for (auto& gr : group_refs) if (e->gpu.ref[gr.first] > gr.second) return 0;
    return !group_refs.empty() && group_refs.size() <= e->cpu.free_ids.size() ? 1 : 0;
}
int32_t mi355_be_swap_in_required_blocks(void* be, const int64_t* seq_ids, int32_t n) {             // :1108-1120
    Engine* e = E(be);
    std::unordered_set<int> ids;
    for (int i = 0; i < n; ++i) {
        auto it = e->tables.find(seq_ids[i]);
        if (it == e->tables.end()) continue;
        for (Blk b : it->second) if (!is_gpu(b)) ids.insert(cpu_id(b));
    }
    return (int32_t)ids.size();
}
int32_t mi355_be_can_swap_in(void* be, const int64_t* seq_ids, int32_t n) {
    return mi355_be_swap_in_required_blocks(be, seq_ids, n) <= (int32_t)E(be)->gpu.free_ids.size() ? 1 : 0;
}
static int32_t emit_pairs(const std::vector<std::pair<int, int>>& m, int64_t* pairs, int32_t cap) {
    int k = 0;
    for (auto& p : m) { if (k < cap) { pairs[2 * k] = p.first; pairs[2 * k + 1] = p.second; } ++k; }
    return k;
}
int32_t mi355_be_swap_out(void* be, int64_t group_id, const int64_t* seq_ids, int32_t n, int64_t* pairs, int32_t cap) {  // :1122-1177
    Engine* e = E(be);
    auto pc = prefix_block_counts(e, seq_ids, n);
    std::unordered_map<int, int> map;                     // gpu id -> cpu id
    std::vector<std::pair<int, int>> order;
    Pending pend;
    if (e->refuse_swap_out > 0) { --e->refuse_swap_out; return -4; }
    {   // every failure is detected BEFORE a table or a pool is touched (no half-swapped group)
        std::unordered_set<int> need;
        for (int i = 0; i < n; ++i) {
            auto it = e->tables.find(seq_ids[i]);
            if (it == e->tables.end()) return -1;
            for (size_t k = pc[seq_ids[i]]; k < it->second.size(); ++k) need.insert(it->second[k]);
        }
        if (need.size() > e->cpu.free_ids.size()) return -2;
        if ((int64_t)need.size() > (int64_t)cap) return -3;       // the caller's pair buffer is too small
    }
    for (int i = 0; i < n; ++i) {
        auto it = e->tables.find(seq_ids[i]);
        if (it == e->tables.end()) return -1;
        std::vector<Blk> old = it->second, nt = it->second;
        for (size_t k = pc[seq_ids[i]]; k < old.size(); ++k) {
            const int g = old[k];
            auto f = map.find(g);
            int c;
            if (f == map.end()) {
                if (e->cpu.free_ids.empty()) return -2;
                c = e->cpu.allocate(); map[g] = c; order.push_back({g, c});
            } else { c = f->second; ++e->cpu.ref[c]; }
            nt[k] = cpu_code(c);
        }
        pend.old_tables.push_back({seq_ids[i], old});
        pend.new_tables.push_back({seq_ids[i], nt});
        it->second = nt;
    }
    for (size_t t = 0; t < pend.old_tables.size(); ++t)
        for (Blk ob : pend.old_tables[t].second)
            if (std::find(pend.new_tables[t].second.begin(), pend.new_tables[t].second.end(), ob) == pend.new_tables[t].second.end())
                e->gpu.release(ob);
    e->pending_out[group_id] = std::move(pend);
    return emit_pairs(order, pairs, cap);
}
int32_t mi355_be_swap_in(void* be, int64_t group_id, const int64_t* seq_ids, int32_t n, int64_t* pairs, int32_t cap) {   // :1223-1264
    Engine* e = E(be);
    std::unordered_map<int, int> map;                     // cpu id -> gpu id
    std::vector<std::pair<int, int>> order;
    Pending pend;
    if (e->refuse_swap_in > 0) { --e->refuse_swap_in; return -4; }
    {   // check everything before mutating, as in swap_out
        const int32_t need = mi355_be_swap_in_required_blocks(be, seq_ids, n);
        for (int i = 0; i < n; ++i) if (e->tables.find(seq_ids[i]) == e->tables.end()) return -1;
        if (need > (int32_t)e->gpu.free_ids.size()) return -2;
        if (need > cap) return -3;
    }
    for (int i = 0; i < n; ++i) {
        auto it = e->tables.find(seq_ids[i]);
        if (it == e->tables.end()) return -1;
        std::vector<Blk> old = it->second, nt = it->second;
        for (size_t k = 0; k < old.size(); ++k) {
            if (is_gpu(old[k])) continue;
            const int c = cpu_id(old[k]);
            auto f = map.find(c);
            int g;
            if (f == map.end()) {
                if (e->gpu.free_ids.empty()) return -2;
                g = e->gpu.allocate(); map[c] = g; order.push_back({c, g});
            } else { g = f->second; ++e->gpu.ref[g]; }
            nt[k] = g;
        }
        pend.old_tables.push_back({seq_ids[i], old});
        pend.new_tables.push_back({seq_ids[i], nt});
        it->second = nt;
    }
    e->pending_in[group_id] = std::move(pend);
    return emit_pairs(order, pairs, cap);
}
// TEST HOOK: the next `n_out` calls of mi355_be_swap_out and the next `n_in` calls of mi355_be_swap_in return -4 before they
// touch a table or a pool -- the only way to reach the scheduler's refused-swap branches once `can_swap_*` has said yes.
void mi355_be_test_refuse_swaps(void* be, int32_t n_out, int32_t n_in) { E(be)->refuse_swap_out = n_out; E(be)->refuse_swap_in = n_in; }
void mi355_be_finalize_swap_out(void* be, int64_t group_id) { E(be)->pending_out.erase(group_id); }
void mi355_be_rollback_swap_out(void* be, int64_t group_id) {                                       // :1270-1295
    Engine* e = E(be);
    auto it = e->pending_out.find(group_id);
    if (it == e->pending_out.end()) return;
    Pending p = std::move(it->second); e->pending_out.erase(it);
    for (size_t t = 0; t < p.old_tables.size(); ++t) {
        auto& ot = p.old_tables[t].second; auto& nt = p.new_tables[t].second;
        for (Blk nb : nt) if (std::find(ot.begin(), ot.end(), nb) == ot.end()) e->release(nb);
        for (Blk ob : ot) if (std::find(nt.begin(), nt.end(), ob) == nt.end()) e->retain(ob);
        e->tables[p.old_tables[t].first] = ot;
    }
}
void mi355_be_finalize_swap_in(void* be, int64_t group_id) {                                        // :1297-1311
    Engine* e = E(be);
    auto it = e->pending_in.find(group_id);
    if (it == e->pending_in.end()) return;
    Pending p = std::move(it->second); e->pending_in.erase(it);
    for (size_t t = 0; t < p.old_tables.size(); ++t) {
        auto& ot = p.old_tables[t].second; auto& nt = p.new_tables[t].second;
        for (Blk ob : ot) if (std::find(nt.begin(), nt.end(), ob) == nt.end()) e->release(ob);
    }
}
void mi355_be_rollback_swap_in(void* be, int64_t group_id) {                                        // :1313-1329
    Engine* e = E(be);
    auto it = e->pending_in.find(group_id);
    if (it == e->pending_in.end()) return;
    Pending p = std::move(it->second); e->pending_in.erase(it);
    for (auto& ot : p.old_tables) {
        auto cur = e->tables.find(ot.first);
        if (cur != e->tables.end())
            for (Blk nb : cur->second) if (std::find(ot.second.begin(), ot.second.end(), nb) == ot.second.end()) e->release(nb);
        e->tables[ot.first] = ot.second;
    }
}

/* ---- bare PrefixCache (prefix_cache.rs tests drive it without an engine): blocks are plain ids with external refs */
void* mi355_pc_create(int32_t block_size, int32_t enabled, int32_t max_cached_blocks, int32_t num_block_ids) {
    Engine* e = new Engine();
    e->block_size = block_size; e->has_cache = true;
    e->cache.block_size = block_size; e->cache.cfg_enabled = enabled != 0; e->cache.max_blocks = (size_t)std::max(0, max_cached_blocks);
    e->gpu.ref.assign(std::max(1, num_block_ids), 1);     /* block(id) fixtures start with refcount 1 (:392-399) */
    return e;
}
int32_t mi355_pc_insert(void* pc, const uint32_t* tokens, int32_t n, const int32_t* blocks, int32_t nblocks, int32_t* evicted, int32_t cap) {
    Engine* e = E(pc);
    auto ev = e->cache.insert(tokens, n, blocks, nblocks, e->gpu.ref, false, 0, 0);
    for (size_t i = 0; i < ev.size() && (int)i < cap; ++i) evicted[i] = ev[i];
    return (int32_t)ev.size();
}
int32_t mi355_pc_match(void* pc, const uint32_t* tokens, int32_t n, int32_t* blocks, int32_t cap) {
    Engine* e = E(pc);
    uint64_t lh = 0;
    const int m = e->cache.match(tokens, n, false, 0, 0, &lh);
    if (m > 0 && blocks) { auto b = e->cache.blocks_for_match(lh); for (size_t i = 0; i < b.size() && (int)i < cap; ++i) blocks[i] = b[i]; }
    return m;
}
int32_t mi355_pc_evict(void* pc, int32_t num, const uint32_t* protect_tokens, int32_t protect_n, int32_t* evicted, int32_t cap) {
    Engine* e = E(pc);
    std::unordered_set<uint64_t> prot;
    if (protect_tokens && protect_n > 0) { uint64_t lh = 0; if (e->cache.match(protect_tokens, protect_n, false, 0, 0, &lh) > 0) prot.insert(lh); }
    auto ev = e->cache.evict((size_t)num, prot);
    for (size_t i = 0; i < ev.size() && (int)i < cap; ++i) evicted[i] = ev[i];
    return (int32_t)ev.size();
}
int32_t mi355_pc_cached_blocks(void* pc) { return (int32_t)E(pc)->cache.entries.size(); }
int32_t mi355_pc_lru_len(void* pc) { return (int32_t)E(pc)->cache.leaf_lru.size(); }
uint64_t mi355_pc_hash_for_blocks(void* pc, const uint32_t* tokens, int32_t n, int32_t full_blocks, int32_t has_seed, uint64_t seed, int32_t seed_block) {
    uint64_t h = 0;
    return E(pc)->cache.hash_for_blocks(tokens, n, full_blocks, has_seed != 0, seed, seed_block, &h) ? h : 0;
}

/* ---- a1 / a2: InputMetadata arrays from the engine state (inputs.rs:90-230, 376-454).  HOST outputs.
 * decode: tokens/positions/slots/context_lens [n]; block_tables [n, max_blocks] zero padded, rows truncated to
 * ceil(len/bs) (:425-430).  Returns max_blocks (the padded width), or < 0. */
int32_t mi355_be_prepare_decode(void* be, const int64_t* seq_ids, int32_t n, uint32_t* tokens, int64_t* positions,
                                int64_t* slot_mapping, uint32_t* context_lens, uint32_t* block_tables, int32_t bt_cap_cols) {
    Engine* e = E(be);
    int maxb = 0;
    for (int i = 0; i < n; ++i) {
        Seq* s = find_seq(e, seq_ids[i]);
        auto it = e->tables.find(seq_ids[i]);
        if (!s || it == e->tables.end() || s->tokens.empty()) return -1;
        const int len = (int)s->tokens.size(), pos = len - 1;
        const int used = (len + e->block_size - 1) / e->block_size;
        if (pos / e->block_size >= (int)it->second.size()) return -2;      /* "Block table is too small" */
        maxb = std::max(maxb, std::min(used, (int)it->second.size()));
        tokens[i] = s->tokens.back(); positions[i] = pos; context_lens[i] = (uint32_t)len;
        const Blk b = it->second[pos / e->block_size];
        if (!is_gpu(b)) return -3;
        slot_mapping[i] = (int64_t)b * e->block_size + pos % e->block_size;
    }
    if (maxb > bt_cap_cols) return -4;
    for (int i = 0; i < n; ++i) {
        Seq* s = find_seq(e, seq_ids[i]);
        auto& t = e->tables[seq_ids[i]];
        const int used = std::min(((int)s->tokens.size() + e->block_size - 1) / e->block_size, (int)t.size());
        for (int k = 0; k < maxb; ++k) block_tables[(size_t)i * maxb + k] = k < used ? (uint32_t)t[k] : 0u;
    }
    return maxb;
}
/* prompt step: tokens after num_cached, up to `chunk` of them (0 = all).  Outputs sized by the caller:
 * tokens/positions/slots [sum chunk]; context_lens [n]; cu_seqlens_q/k [n+1]; block_tables [n, max_blocks].
 * Returns the number of tokens; *max_blocks_out = padded table width. */
int32_t mi355_be_prepare_prompt(void* be, const int64_t* seq_ids, int32_t n, int32_t chunk, uint32_t* tokens,
                                int64_t* positions, int64_t* slot_mapping, uint32_t* context_lens, uint32_t* cu_q,
                                uint32_t* cu_k, uint32_t* block_tables, int32_t tok_cap, int32_t bt_cap_cols,
                                int32_t* max_blocks_out) {
    Engine* e = E(be);
    int T = 0, maxb = 0;
    cu_q[0] = 0; cu_k[0] = 0;
    for (int i = 0; i < n; ++i) {
        Seq* s = find_seq(e, seq_ids[i]);
        auto it = e->tables.find(seq_ids[i]);
        if (!s || it == e->tables.end()) return -1;
        const int cached = s->num_cached;
        const int take = e->prefill_chunk_tokens(*s, chunk);
        if (T + take > tok_cap) return -4;
        for (int j = 0; j < take; ++j) {
            const int pos = cached + j;
            tokens[T] = s->tokens[pos]; positions[T] = pos;
            if (pos / e->block_size >= (int)it->second.size()) slot_mapping[T] = -1;           /* _PAD_SLOT_ID */
            else slot_mapping[T] = (int64_t)it->second[pos / e->block_size] * e->block_size + pos % e->block_size;
            ++T;
        }
        cu_q[i + 1] = cu_q[i] + take; cu_k[i + 1] = cu_k[i] + cached + take;
        context_lens[i] = (uint32_t)(cached + take);
        maxb = std::max(maxb, (int)it->second.size());
    }
    if (maxb > bt_cap_cols) return -4;
    for (int i = 0; i < n; ++i) {
        auto& t = e->tables[seq_ids[i]];
        for (int k = 0; k < maxb; ++k) block_tables[(size_t)i * maxb + k] = k < (int)t.size() && is_gpu(t[k]) ? (uint32_t)t[k] : 0u;
    }
    if (max_blocks_out) *max_blocks_out = maxb;
    return T;
}

}  // extern "C"

// =====================================================================================================================
// Continuous-batching scheduler (src/scheduler/mod.rs:66-839): three queues (waiting / running / swapped), prompt
// steps and decode steps alternate (`is_last_prefill`), per-step prefill token budget, chunked prefill re-queueing,
// preemption of the latest arrivals by recompute (no prefix cache, single sequence) or swap, FCFS priorities.
// Wall-clock inputs of the reference (arrival time, swap cooling period of 300 ms, mod.rs:41) are explicit arguments
// here so the policy is deterministic and testable.
// =====================================================================================================================
namespace {

enum GroupStatus { G_WAITING = 0, G_PENDING = 1, G_RUNNING = 2, G_SWAPPED = 3, G_FINISHED = 4, G_ABORTED = 5, G_IGNORED = 6 };
enum { R_SCHEDULED = 0, R_IGNORED = 1, R_SWAP_IN_PAIRS = 2, R_SWAP_OUT_PAIRS = 3, R_COPY_PAIRS = 4, R_SWAP_IN_GROUPS = 5,
       R_SWAP_OUT_GROUPS = 6, R_RUNNER_RELEASES = 7, R_COUNT = 8 };

struct Group { std::vector<int64_t> seqs; uint64_t arrival = 0; int status = G_WAITING; bool has_swapped_time = false; uint64_t swapped_ms = 0; };

struct Sched {
    Engine* eng = nullptr;
    std::deque<int64_t> waiting, running, swapped;
    std::unordered_map<int64_t, Group> groups;
    int max_parallel = 1, max_batched_tokens = 1, chunk = 0;
    bool is_last_prefill = false;
    std::vector<int64_t> result[R_COUNT];
    std::vector<int64_t> pending_releases;

    Group& g(int64_t id) { return groups[id]; }
    int group_prefill_tokens(const Group& gr) { int t = 0; for (int64_t s : gr.seqs) { Seq* q = find_seq(eng, s); if (q) t += eng->prefill_chunk_tokens(*q, chunk); } return t; }
    bool has_block_table(const Group& gr) { for (int64_t s : gr.seqs) if (!eng->tables.count(s)) return false; return true; }
    void free_group(const Group& gr, bool cache_prefix) {                       // _free (mod.rs:766-775)
        for (int64_t s : gr.seqs) {
            if (cache_prefix && gr.status == G_FINISHED) mi355_be_cache_sequence(eng, s);
            mi355_be_free_sequence(eng, s);
        }
    }
    void request_release(const Group& gr) { for (int64_t s : gr.seqs) if (std::find(pending_releases.begin(), pending_releases.end(), s) == pending_releases.end()) pending_releases.push_back(s); }
    void remove_everywhere(int64_t gid) {
        for (auto* q : {&waiting, &running, &swapped}) { auto it = std::find(q->begin(), q->end(), gid); if (it != q->end()) q->erase(it); }
    }
    void abort_group(int64_t gid) { remove_everywhere(gid); Group& gr = g(gid); gr.status = G_ABORTED; free_group(gr, false); }
    void preempt_by_recompute(int64_t gid) {                                    // :718-723
        Group& gr = g(gid);
        request_release(gr);
        gr.status = G_WAITING;
        free_group(gr, false);
        waiting.push_front(gid);
    }
    void preempt(int64_t gid, uint64_t now_ms) {                                // :700-764
        Group& gr = g(gid);
        if (gr.seqs.size() == 1 && !eng->has_cache) { preempt_by_recompute(gid); return; }
        if (!mi355_be_can_swap_out(eng, gr.seqs.data(), (int)gr.seqs.size())) {
            if (gr.seqs.size() == 1) { preempt_by_recompute(gid); return; }
            request_release(gr);
            abort_group(gid);
            return;
        }
        size_t nblk = 0;                                                        // upper bound of the pair count: every block of the group
        for (int64_t sid : gr.seqs) { auto it = eng->tables.find(sid); if (it != eng->tables.end()) nblk += it->second.size(); }
        std::vector<int64_t> pairs(2 * nblk + 2);
        const int k = mi355_be_swap_out(eng, gid, gr.seqs.data(), (int)gr.seqs.size(), pairs.data(), (int32_t)nblk);
        if (k < 0) {
            // the swap was refused by its up-front checks (nothing moved): the group must not be recorded as
            // swapped out.  Fall back the way `can_swap_out == false` does: recompute a single sequence, abort a larger group.
            if (gr.seqs.size() == 1) { preempt_by_recompute(gid); return; }
            request_release(gr);
            abort_group(gid);
            return;
        }
        for (int i = 0; i < k; ++i) { result[R_SWAP_OUT_PAIRS].push_back(pairs[2 * i]); result[R_SWAP_OUT_PAIRS].push_back(pairs[2 * i + 1]); }
        result[R_SWAP_OUT_GROUPS].push_back(gid);
        gr.has_swapped_time = true; gr.swapped_ms = now_ms;
        gr.status = G_SWAPPED;
        swapped.push_back(gid);
    }
    void append_token_slots(const Group& gr) {                                  // :680-698
        for (int64_t s : gr.seqs) {
            int32_t src = -1, dst = -1;
            if (mi355_be_append_token_slot(eng, s, &src, &dst) == 1) { result[R_COPY_PAIRS].push_back(src); result[R_COPY_PAIRS].push_back(dst); }
        }
    }
    bool ensure_prefill_chunk_slots(const Group& gr) {                          // :814-838
        const int req = mi355_be_prefill_chunk_blocks_required(eng, gr.seqs.data(), (int)gr.seqs.size(), chunk);
        if (req > (int)eng->gpu.free_ids.size()) eng->evict_until_free(req);
        if (mi355_be_can_append_prefill_chunk(eng, gr.seqs.data(), (int)gr.seqs.size(), chunk) != 1) return false;
        mi355_be_append_prefill_chunk_slots(eng, gr.seqs.data(), (int)gr.seqs.size(), chunk);
        return true;
    }
    int evict_under_pressure() {                                                // :804-812 (10 % of the cache, at least one)
        const int cached = eng->has_cache ? (int)eng->cache.entries.size() : 0;
        if (cached == 0) return 0;
        int blocks = (int)((cached + 9) / 10);
        if (blocks < 1) blocks = 1;
        return mi355_be_evict_prefix_cache_blocks(eng, blocks);
    }
    void sort_fcfs(std::deque<int64_t>& q) {                                    // earliest arrival LAST (:777-789)
        std::stable_sort(q.begin(), q.end(), [&](int64_t a, int64_t b) { return groups[a].arrival < groups[b].arrival; });
        std::reverse(q.begin(), q.end());
    }
};

Sched* S(void* p) { return static_cast<Sched*>(p); }

}  // namespace

extern "C" {

void* mi355_sched_create(int32_t block_size, int32_t num_gpu_blocks, int32_t num_cpu_blocks, int32_t prefix_cache_enabled,
                         int32_t max_cached_blocks, int32_t max_num_parallel_reqs, int32_t max_num_batched_tokens,
                         int32_t prefill_chunk_size) {
    void* be = mi355_be_create(block_size, num_gpu_blocks, num_cpu_blocks, prefix_cache_enabled, max_cached_blocks);
    if (!be) return nullptr;
    Sched* s = new Sched();
    s->eng = E(be);
    s->max_parallel = std::max(1, max_num_parallel_reqs);
    s->max_batched_tokens = std::max(1, max_num_batched_tokens);
    s->chunk = std::max(0, prefill_chunk_size);
    return s;
}
void mi355_sched_destroy(void* sp) { Sched* s = S(sp); if (!s) return; delete s->eng; delete s; }
void* mi355_sched_block_engine(void* sp) { return S(sp)->eng; }

/* Scheduler::add_sequence (mod.rs:123-125); the sequences were created with mi355_be_seq_create on the engine */
int32_t mi355_sched_add_group(void* sp, int64_t group_id, const int64_t* seq_ids, int32_t n, uint64_t arrival) {
    Sched* s = S(sp);
    if (n < 1 || s->groups.count(group_id)) return -1;
    for (int i = 0; i < n; ++i) if (!find_seq(s->eng, seq_ids[i])) return -1;
    Group gr; gr.seqs.assign(seq_ids, seq_ids + n); gr.arrival = arrival;
    s->groups[group_id] = gr;
    s->waiting.push_back(group_id);
    return 0;
}
int32_t mi355_sched_group_status(void* sp, int64_t group_id) { auto it = S(sp)->groups.find(group_id); return it == S(sp)->groups.end() ? -1 : it->second.status; }
int32_t mi355_sched_set_group_finished(void* sp, int64_t group_id) { auto it = S(sp)->groups.find(group_id); if (it == S(sp)->groups.end()) return -1; it->second.status = G_FINISHED; return 0; }
int32_t mi355_sched_queue_len(void* sp, int32_t which) { Sched* s = S(sp); return (int32_t)(which == 0 ? s->waiting.size() : which == 1 ? s->running.size() : s->swapped.size()); }
int32_t mi355_sched_has_unfinished(void* sp) { Sched* s = S(sp); return (!s->running.empty() || !s->waiting.empty()) ? 1 : 0; }
int32_t mi355_sched_is_last_prefill(void* sp) { return S(sp)->is_last_prefill ? 1 : 0; }
/* results of the last schedule() / take_pending_runner_releases(): which = R_* above; pairs come flattened */
int32_t mi355_sched_result(void* sp, int32_t which, int64_t* out, int32_t cap) {
    Sched* s = S(sp);
    if (which < 0 || which >= R_COUNT) return -1;
    if (which == R_RUNNER_RELEASES) { s->result[which] = s->pending_releases; s->pending_releases.clear(); }
    const auto& v = s->result[which];
    for (size_t i = 0; i < v.size() && (int)i < cap; ++i) out[i] = v[i];
    return (int32_t)v.size();
}

/* Scheduler::schedule (mod.rs:183-455).  Returns 1 for a PROMPT step (scheduled = newly admitted groups), 0 for a
 * DECODE step (scheduled = the running queue).  now_ms feeds the swap cooling period. */
int32_t mi355_sched_schedule(void* sp, uint64_t now_ms) {
    Sched* s = S(sp);
    Engine* e = s->eng;
    for (int i = 0; i < R_RUNNER_RELEASES; ++i) s->result[i].clear();
    if (s->swapped.empty()) {
        const size_t pre_existing_running = s->running.size();
        int scheduled_tokens = 0;
        while (!s->waiting.empty()) {
            if (s->is_last_prefill && pre_existing_running > 0) break;             // interleave prompt and decode steps
            const int64_t gid = s->waiting.front();
            Group& gr = s->g(gid);
            const int group_tokens = s->group_prefill_tokens(gr);
            if (group_tokens > 0 && scheduled_tokens + group_tokens > s->max_batched_tokens) break;
            if ((int)s->running.size() >= s->max_parallel) break;
            size_t individual = 0;
            for (int64_t rg : s->running) individual += s->g(rg).seqs.size();
            if ((int)individual + 1 > s->max_parallel) break;
            if (!s->has_block_table(gr)) {
                const int st = mi355_be_can_allocate(e, gr.seqs.data(), (int)gr.seqs.size(), s->chunk);
                if (st == 1) break;                                                // Later
                if (st == 2) {                                                     // Impossible: prompt exceeds the pool
                    gr.status = G_IGNORED;
                    s->result[R_IGNORED].push_back(gid);
                    s->waiting.pop_front();
                    continue;
                }
                if (mi355_be_allocate(e, gr.seqs.data(), (int)gr.seqs.size(), s->chunk) != 0) break;
            } else if (!s->ensure_prefill_chunk_slots(gr)) {
                break;
            }
            gr.status = G_RUNNING;
            s->waiting.pop_front();
            s->running.push_back(gid);
            s->result[R_SCHEDULED].push_back(gid);
            scheduled_tokens += group_tokens;
        }
        if (!s->result[R_SCHEDULED].empty() || !s->result[R_IGNORED].empty()) { s->is_last_prefill = true; return 1; }
    }
    // ---- decode step: reserve a token slot for every running group, preempting the latest arrivals when out of blocks
    s->sort_fcfs(s->running);
    std::deque<int64_t> running;
    bool any_preempted = false;
    while (!s->running.empty()) {
        if ((int)running.size() >= s->max_parallel) {
            while (!s->running.empty()) { const int64_t ex = s->running.front(); s->running.pop_front(); s->preempt(ex, now_ms); any_preempted = true; }
            break;
        }
        const int64_t gid = s->running.front(); s->running.pop_front();
        bool self_preempted = false;
        while (mi355_be_can_append_token(e, s->g(gid).seqs.data(), (int)s->g(gid).seqs.size()) != 1) {
            if (s->evict_under_pressure() > 0) continue;
            if (!s->running.empty()) {
                const int64_t victim = s->running.back(); s->running.pop_back();
                s->preempt(victim, now_ms); any_preempted = true;
            } else {
                s->preempt(gid, now_ms); any_preempted = true; self_preempted = true;
                break;
            }
        }
        if (!self_preempted) { s->append_token_slots(s->g(gid)); running.push_back(gid); }
    }
    s->running = running;
    s->sort_fcfs(s->swapped);
    if (!any_preempted) {
        while (!s->swapped.empty()) {
            const int64_t gid = s->swapped.front();
            Group& gr = s->g(gid);
            if (gr.has_swapped_time && now_ms - gr.swapped_ms < 300) break;          // SWAP_COOLING_PERIOD
            // room for the swapped blocks AND for the slot that is reserved right after the swap-in (mod.rs:395-396).
            // The reference checks only the former (`can_swap_in_seq_group`, block_engine.rs:1214-1219); when the group
            // then needs one more block than the pool has, its allocator unwraps an empty free list (:113-117) and
            // panics.  Here the group stays swapped out until both fit (found by the block-pressure stress test).
            int need = mi355_be_swap_in_required_blocks(e, gr.seqs.data(), (int)gr.seqs.size());
            for (int64_t sid : gr.seqs) { Seq* q = find_seq(e, sid); if (q) need += blocks_missing_for_sequence(e, sid, *q); }
            if (need > (int)e->gpu.free_ids.size()) {
                e->evict_until_free(need);
                if (need > (int)e->gpu.free_ids.size()) break;
            }
            s->swapped.pop_front();
            size_t nblk = 0;
            for (int64_t sid : gr.seqs) { auto it = e->tables.find(sid); if (it != e->tables.end()) nblk += it->second.size(); }
            std::vector<int64_t> pairs(2 * nblk + 2);
            const int k = mi355_be_swap_in(e, gid, gr.seqs.data(), (int)gr.seqs.size(), pairs.data(), (int32_t)nblk);
            if (k < 0) { s->swapped.push_front(gid); break; }                        // refused, nothing moved: the group stays swapped out, first in line
            for (int i = 0; i < k; ++i) { s->result[R_SWAP_IN_PAIRS].push_back(pairs[2 * i]); s->result[R_SWAP_IN_PAIRS].push_back(pairs[2 * i + 1]); }
            s->result[R_SWAP_IN_GROUPS].push_back(gid);
            gr.has_swapped_time = false;
            gr.status = G_RUNNING;
            s->append_token_slots(gr);
            s->running.push_back(gid);
        }
    }
    s->is_last_prefill = false;
    for (int64_t gid : s->running) s->result[R_SCHEDULED].push_back(gid);
    return 0;
}

/* filter_prefill_finished (mod.rs:542-616): after a prompt step, groups whose prompt is not finished advance
 * num_cached_tokens by the chunk and go back to `waiting`; returns the count of FINISHED groups (ids in out). */
int32_t mi355_sched_filter_prefill_finished(void* sp, const int64_t* scheduled, int32_t n, int64_t* finished_out, int32_t cap) {
    Sched* s = S(sp);
    if (s->chunk <= 0) return -1;
    int k = 0;
    for (int i = 0; i < n; ++i) {
        auto it = s->groups.find(scheduled[i]);
        if (it == s->groups.end()) return -1;
        Seq* q = find_seq(s->eng, it->second.seqs[0]);
        const int take = s->eng->prefill_chunk_tokens(*q, s->chunk);
        if (take == 0 || q->num_cached + take >= q->prompt_len) {
            if (k < cap) finished_out[k] = scheduled[i];
            ++k;
        } else {
            q->num_cached += take;
            if (!(q->has_warmup && q->warmup > q->num_cached && q->warmup < q->prompt_len)) q->has_warmup = false;
            it->second.status = G_PENDING;
            s->waiting.push_back(scheduled[i]);
            auto r = std::find(s->running.begin(), s->running.end(), scheduled[i]);
            if (r != s->running.end()) s->running.erase(r);
        }
    }
    return k;
}
/* free_finished_sequence_groups (mod.rs:465-504): returns the released sequence ids */
int32_t mi355_sched_free_finished(void* sp, int64_t* released_out, int32_t cap) {
    Sched* s = S(sp);
    std::deque<int64_t> keep; std::vector<int64_t> fin;
    for (int64_t gid : s->running) (s->g(gid).status == G_FINISHED ? (void)fin.push_back(gid) : (void)keep.push_back(gid));
    s->running = keep;
    int k = 0;
    for (int64_t gid : fin) {
        Group& gr = s->g(gid);
        for (int64_t q : gr.seqs) { if (k < cap) released_out[k] = q; ++k; }
        s->free_group(gr, true);
    }
    return k;
}
/* abort_sequences (mod.rs:618-657): the whole group of every named sequence is removed and freed */
int32_t mi355_sched_abort_sequences(void* sp, const int64_t* seq_ids, int32_t n) {
    Sched* s = S(sp);
    std::vector<int64_t> gids;
    for (auto& kv : s->groups) {
        if (kv.second.status == G_ABORTED || kv.second.status == G_IGNORED) continue;
        bool queued = false;
        for (auto* q : {&s->waiting, &s->running, &s->swapped}) if (std::find(q->begin(), q->end(), kv.first) != q->end()) queued = true;
        if (!queued) continue;
        for (int64_t q : kv.second.seqs) if (std::find(seq_ids, seq_ids + n, q) != seq_ids + n) { gids.push_back(kv.first); break; }
    }
    for (int64_t gid : gids) {
        Group& gr = s->g(gid);
        const bool has_table = s->has_block_table(gr);
        s->remove_everywhere(gid);
        gr.status = G_ABORTED;
        if (has_table) s->free_group(gr, false);
    }
    return (int32_t)gids.size();
}
/* rollback_swap_in_groups / rollback_swap_out_groups (mod.rs:146-181) + the engine-side table rollback */
void mi355_sched_rollback_swap_in(void* sp, int64_t group_id) {
    Sched* s = S(sp);
    auto it = std::find(s->running.begin(), s->running.end(), group_id);
    if (it == s->running.end()) return;
    s->running.erase(it);
    s->g(group_id).status = G_SWAPPED;
    s->swapped.push_front(group_id);
    mi355_be_rollback_swap_in(s->eng, group_id);
}
void mi355_sched_rollback_swap_out(void* sp, int64_t group_id) {
    Sched* s = S(sp);
    auto it = std::find(s->swapped.begin(), s->swapped.end(), group_id);
    if (it == s->swapped.end()) return;
    s->swapped.erase(it);
    Group& gr = s->g(group_id);
    gr.status = G_RUNNING; gr.has_swapped_time = false;
    s->running.push_front(group_id);
    mi355_be_rollback_swap_out(s->eng, group_id);
}

/* ---- cache budget: src/lib.rs:128-284,515-523 -------------------------------------------------------- */
/* compute_kvcache_budget_bytes (lib.rs:515-523): round(free_bytes * fraction) with the f32 fraction widened to f64;
 * -1 when the fraction is outside (0, 1] (the reference returns an error) */
int64_t mi355_kvcache_budget_bytes(int64_t free_bytes, float fraction) {
    if (!(0.0f < fraction && fraction <= 1.0f) || free_bytes < 0) return -1;
    const double v = (double)free_bytes * (double)fraction;
    return (int64_t)(v + 0.5);                                  /* f64::round for non-negative values */
}
/* get_cache_config (lib.rs:128-284), the non-MLA / non-TurboQuant arm: per_block = dsize * block_size * kv heads per
 * shard * head_dim * layers * 2 (at least 1); num_gpu_blocks = mem_gpu_mb * 2^20 / per_block; CPU blocks: half the GPU
 * blocks when mem_cpu_mb == 0, else the explicit budget -- and none at all when CPU swap is off (the reference ties
 * that to its `cuda` feature, lib.rs:247-257).  kv heads per shard: global / shards, 1 when there are fewer heads than
 * shards (:159-166).  Returns 0, or hipErrorInvalidValue (1) for non-positive dimensions. */
int mi355_get_cache_config(int64_t mem_gpu_mb, int64_t mem_cpu_mb, int32_t block_size, int32_t num_kv_heads, int32_t head_dim,
                           int32_t num_layers, int32_t dsize, int32_t num_shards, int32_t cpu_swap, int64_t* num_gpu_blocks,
                           int64_t* num_cpu_blocks) {
    if (mem_gpu_mb < 0 || mem_cpu_mb < 0 || block_size <= 0 || num_kv_heads <= 0 || head_dim <= 0 || dsize <= 0 ||
        !num_gpu_blocks || !num_cpu_blocks || mem_gpu_mb > (INT64_MAX >> 20) || mem_cpu_mb > (INT64_MAX >> 20))
        return 1;
    const int64_t shards = num_shards > 1 ? num_shards : 1;
    const int64_t heads = num_kv_heads < shards ? 1 : num_kv_heads / shards;
    const int64_t layers = num_layers > 1 ? num_layers : 1;     /* kv_cache_num_layers().max(1) */
    int64_t per_block = (int64_t)dsize * block_size * heads * head_dim * layers * 2;
    if (per_block < 1) per_block = 1;
    const int64_t gpu = mem_gpu_mb * (1 << 20) / per_block;
    *num_gpu_blocks = gpu;
    *num_cpu_blocks = cpu_swap ? (mem_cpu_mb == 0 ? gpu / 2 : mem_cpu_mb * (1 << 20) / per_block) : 0;
    return 0;
}


}  // extern "C"
