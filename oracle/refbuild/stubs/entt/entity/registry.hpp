// Stand-in for EnTT (skypjack/entt, v3.10-era API; the reference vendors it as an empty git submodule, ext/entt).
// Test infrastructure only: it lets the reference's own physics sources (oracle/refbuild/build_ref.py) compile and run
// here so that oracle/ can be pinned against the original code.  Written from EnTT's documented behaviour, not from
// its sources; only the API surface the reference's physics-only build touches is provided.
//
// The properties of EnTT that the reference's results depend on, and that this file therefore keeps:
//   * a component pool is a packed array: emplace appends, erase is swap-with-last-and-pop;
//   * views and groups iterate a pool from its LAST element to its first (hence the reference's
//     `numColliders - 1 - index` arithmetic);
//   * an owning group keeps the entities that have all of its components at the FRONT of every owned pool, in the order
//     in which they entered the group (swap into position `length` on entry, swap with position `length - 1` on exit),
//     nested groups (one group's constraint set containing another's) included;
//   * entity identifiers are handed out sequentially, destroyed identifiers are recycled LIFO with a bumped version
//     (12 version bits above 20 index bits);
//   * storage<T>::raw() is an array of pages; the pool is one contiguous, address-stable allocation here, so
//     `*raw()` is valid for every element (EnTT proper pages at 1024 elements).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <tuple>
#include <type_traits>
#include <typeindex>
#include <unordered_map>
#include <utility>
#include <vector>
#include <sys/mman.h>

#ifndef ENTT_ASSERT
#define ENTT_ASSERT(condition, ...) ((void)0)
#endif

namespace entt {

enum class entity : std::uint32_t {};

struct null_t {
    constexpr operator entity() const noexcept { return entity{0xFFFFFu}; }            // all index bits set
    constexpr bool operator==(null_t) const noexcept { return true; }
    constexpr bool operator!=(null_t) const noexcept { return false; }
    constexpr bool operator==(entity e) const noexcept { return (std::uint32_t(e) & 0xFFFFFu) == 0xFFFFFu; }
    constexpr bool operator!=(entity e) const noexcept { return !(*this == e); }
};
constexpr bool operator==(entity e, null_t n) noexcept { return n == e; }
constexpr bool operator!=(entity e, null_t n) noexcept { return n != e; }
inline constexpr null_t null{};

template <typename... T> struct get_t {};
template <typename... T> struct exclude_t {};
template <typename... T> inline constexpr get_t<T...> get{};
template <typename... T> inline constexpr exclude_t<T...> exclude{};

namespace detail {
constexpr std::uint32_t index_mask = 0xFFFFFu;
constexpr std::uint32_t npos = 0xFFFFFFFFu;
inline std::uint32_t idx(entity e) { return std::uint32_t(e) & index_mask; }

struct group_base;

// type-erased packed pool: entity array + sparse index; payload handled by the typed subclass
struct pool_base {
    std::vector<entity> packed;
    std::vector<std::uint32_t> sparse;          // entity index -> position in packed, npos if absent
    std::vector<group_base*> owners;            // owning groups that keep this pool ordered (least restrictive first)
    std::vector<group_base*> observers;         // every group that has to hear about construct / destroy on this pool
    virtual ~pool_base() = default;
    bool contains(entity e) const { auto i = idx(e); return i < sparse.size() && sparse[i] != npos && packed[sparse[i]] == e; }
    std::uint32_t index(entity e) const { return sparse[idx(e)]; }
    std::size_t size() const { return packed.size(); }
    virtual void swap_elements(std::uint32_t a, std::uint32_t b) = 0;
    virtual void pop_last() = 0;
    void swap_positions(std::uint32_t a, std::uint32_t b) {
        if (a == b) return;
        swap_elements(a, b);
        std::swap(packed[a], packed[b]);
        sparse[idx(packed[a])] = a;
        sparse[idx(packed[b])] = b;
    }
    void erase_raw(entity e) {                   // swap-and-pop
        std::uint32_t pos = index(e), last = std::uint32_t(packed.size() - 1);
        swap_positions(pos, last);
        sparse[idx(e)] = npos;
        packed.pop_back();
        pop_last();
    }
};

template <typename T>
struct pool : pool_base {
    static constexpr std::size_t reserve_elems = std::size_t(1) << 21;
    T* data = nullptr;
    T* page0 = nullptr;       // raw() hands out &page0
    pool() {
        void* p = mmap(nullptr, reserve_elems * sizeof(T), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        data = (p == MAP_FAILED) ? nullptr : static_cast<T*>(p);
        page0 = data;
    }
    ~pool() override {
        for (std::size_t i = 0; i < packed.size(); ++i) data[i].~T();
        if (data) munmap(data, reserve_elems * sizeof(T));
    }
    template <typename... A> T& push(entity e, A&&... a) {
        auto i = idx(e);
        if (i >= sparse.size()) sparse.resize(std::size_t(i) + 1, npos);
        sparse[i] = std::uint32_t(packed.size());
        T* slot = data + packed.size();
        if constexpr (std::is_aggregate_v<T>) new (slot) T{std::forward<A>(a)...};
        else new (slot) T(std::forward<A>(a)...);
        packed.push_back(e);
        return *slot;
    }
    void swap_elements(std::uint32_t a, std::uint32_t b) override { using std::swap; T tmp(std::move(data[a])); data[a].~T(); new (data + a) T(std::move(data[b])); data[b].~T(); new (data + b) T(std::move(tmp)); }
    void pop_last() override { data[packed.size()].~T(); }
    // the subset of entt::storage<T> the reference uses
    std::size_t index_of(entity e) const { return index(e); }
    T** raw() { return packed.empty() ? nullptr : &page0; }
    T& element_at(std::size_t i) { return data[i]; }
    const T* cbegin() const { return data; }
    T& get(entity e) { return data[index(e)]; }
};

struct group_base {
    std::vector<pool_base*> owned, required, excluded;   // required = owned + get
    std::uint32_t length = 0;                              // owning groups: members are [0, length) of every owned pool
    std::vector<std::type_index> key;
    bool qualifies(entity e) const {
        for (auto* p : required) if (!p->contains(e)) return false;
        for (auto* p : excluded) if (p->contains(e)) return false;
        return true;
    }
    bool member(entity e) const { return !owned.empty() && owned[0]->contains(e) && owned[0]->index(e) < length; }
    void enter(entity e) {
        if (owned.empty() || member(e) || !qualifies(e)) return;
        for (auto* p : owned) p->swap_positions(p->index(e), length);
        ++length;
    }
    void leave(entity e) {
        if (owned.empty() || !member(e)) return;
        --length;
        for (auto* p : owned) p->swap_positions(p->index(e), length);
    }
};
} // namespace detail

class registry;

template <typename... T>
struct view_t {
    std::tuple<detail::pool<T>*...> pools;
    detail::pool_base* driver() const {
        detail::pool_base* best = nullptr;
        std::apply([&](auto*... p) { ((best = (!best || p->size() < best->size()) ? p : best), ...); }, pools);
        return best;
    }
    struct iterator {
        const view_t* v; detail::pool_base* drv; std::int64_t pos;
        void skip() { while (pos >= 0 && !v->has_all(drv->packed[std::size_t(pos)])) --pos; }
        bool operator!=(const iterator& o) const { return pos != o.pos; }
        iterator& operator++() { --pos; skip(); return *this; }
        std::tuple<entity, T&...> operator*() const { entity e = drv->packed[std::size_t(pos)]; return std::tuple<entity, T&...>(e, std::get<detail::pool<T>*>(v->pools)->get(e)...); }
    };
    struct iterable { iterator b, e; iterator begin() const { return b; } iterator end() const { return e; } };
    bool has_all(entity e) const { return std::apply([&](auto*... p) { return (p->contains(e) && ...); }, pools); }
    iterable each() const { auto* d = driver(); iterator b{this, d, std::int64_t(d->size()) - 1}; b.skip(); return {b, iterator{this, d, -1}}; }
    std::size_t size() const { static_assert(sizeof...(T) == 1, "size() of a multi-component view is not provided"); return std::get<0>(pools)->size(); }
    std::size_t size_hint() const { return driver()->size(); }
    // entity-only iteration (EnTT: begin()/end() over the entities)
    struct eiterator { iterator it; bool operator!=(const eiterator& o) const { return it != o.it; } eiterator& operator++() { ++it; return *this; } entity operator*() const { return it.drv->packed[std::size_t(it.pos)]; } };
    eiterator begin() const { auto r = each(); return {r.b}; }
    eiterator end() const { auto r = each(); return {r.e}; }
};

template <typename Owned, typename Get> struct group_t;
template <typename... O, typename... G>
struct group_t<std::tuple<O...>, std::tuple<G...>> {
    detail::group_base* g;
    std::tuple<detail::pool<O>*..., detail::pool<G>*...> pools;
    struct iterator {
        const group_t* grp; std::int64_t pos;
        bool owning() const { return sizeof...(O) > 0; }
        detail::pool_base* drv() const { return grp->driver(); }
        void skip() { if (!owning()) while (pos >= 0 && !grp->g->qualifies(drv()->packed[std::size_t(pos)])) --pos; }
        bool operator!=(const iterator& o) const { return pos != o.pos; }
        iterator& operator++() { --pos; skip(); return *this; }
        std::tuple<entity, O&..., G&...> operator*() const {
            entity e = drv()->packed[std::size_t(pos)];
            return std::tuple<entity, O&..., G&...>(e, std::get<detail::pool<O>*>(grp->pools)->get(e)..., std::get<detail::pool<G>*>(grp->pools)->get(e)...);
        }
    };
    struct iterable { iterator b, e; iterator begin() const { return b; } iterator end() const { return e; } };
    detail::pool_base* driver() const { return std::get<0>(pools); }
    iterable each() const {
        std::int64_t n = sizeof...(O) > 0 ? std::int64_t(g->length) : std::int64_t(driver()->size());
        iterator b{this, n - 1}; b.skip();
        return {b, iterator{this, -1}};
    }
    std::size_t size() const { std::size_t n = 0; for (auto it = each().b; it.pos >= 0; ++it) ++n; return n; }
};

class context {
    std::unordered_map<std::type_index, std::shared_ptr<void>> vars;
public:
    template <typename T> T* find() { auto it = vars.find(std::type_index(typeid(T))); return it == vars.end() ? nullptr : static_cast<T*>(it->second.get()); }
    template <typename T, typename... A> T& emplace(A&&... a) {
        auto& slot = vars[std::type_index(typeid(T))];
        if (!slot) slot = std::shared_ptr<void>(new T{std::forward<A>(a)...}, [](void* p) { delete static_cast<T*>(p); });
        return *static_cast<T*>(slot.get());
    }
    template <typename T> bool contains() const { return vars.count(std::type_index(typeid(T))) != 0; }
    template <typename T> bool erase() { return vars.erase(std::type_index(typeid(T))) != 0; }
};

class registry {
    std::vector<entity> entities;                 // slot i holds the identifier currently (or last) living at index i
    std::uint32_t free_head = detail::index_mask; // LIFO free list threaded through `entities` (null index = empty)
    std::unordered_map<std::type_index, std::unique_ptr<detail::pool_base>> pools;
    std::vector<std::unique_ptr<detail::group_base>> groups;
    context vars;

    static std::uint32_t version(entity e) { return std::uint32_t(e) >> 20; }
    static entity compose(std::uint32_t index, std::uint32_t ver) { return entity{(index & detail::index_mask) | (ver << 20)}; }

    template <typename T> detail::pool<T>& assure() {
        auto& slot = pools[std::type_index(typeid(T))];
        if (!slot) slot.reset(new detail::pool<T>());
        return static_cast<detail::pool<T>&>(*slot);
    }
    void constructed(detail::pool_base& p, entity e) { for (auto* g : p.observers) g->enter(e); }       // least restrictive group first
    void destroying(detail::pool_base& p, entity e) { for (auto it = p.observers.rbegin(); it != p.observers.rend(); ++it) (*it)->leave(e); }
    void excluded_constructed(detail::pool_base& p, entity e) { (void)p; (void)e; }

public:
    registry() = default;
    registry(const registry&) = delete;
    registry(registry&&) = default;
    registry& operator=(registry&&) = default;

    entity create() {
        if (free_head != detail::index_mask) {
            std::uint32_t i = free_head;
            free_head = detail::idx(entities[i]);
            return entities[i] = compose(i, version(entities[i]));
        }
        entity e = compose(std::uint32_t(entities.size()), 0);
        entities.push_back(e);
        return e;
    }
    entity create(entity hint) {
        std::uint32_t i = detail::idx(hint);
        if (i == detail::index_mask) return create();
        if (i >= entities.size()) {                       // extend; the skipped indices go on the free list
            std::uint32_t first = std::uint32_t(entities.size());
            entities.resize(std::size_t(i) + 1);
            for (std::uint32_t k = first; k < i; ++k) { entities[k] = compose(free_head, 0); free_head = k; }
            return entities[i] = hint;
        }
        if (valid(entities[i]) && detail::idx(entities[i]) == i) return create();
        // unlink slot i from the free list
        std::uint32_t* link = &free_head;
        while (*link != i) link = reinterpret_cast<std::uint32_t*>(&entities[*link]);
        *link = (*link & ~detail::index_mask) | detail::idx(entities[i]);
        return entities[i] = hint;
    }
    bool valid(entity e) const { auto i = detail::idx(e); return i < entities.size() && entities[i] == e; }
    void destroy(entity e) {
        for (auto& kv : pools) if (kv.second->contains(e)) { destroying(*kv.second, e); kv.second->erase_raw(e); }
        std::uint32_t i = detail::idx(e);
        entities[i] = compose(free_head, (version(e) + 1) & 0xFFFu);
        free_head = i;
    }

    template <typename T, typename... A> decltype(auto) emplace(entity e, A&&... a) {
        auto& p = assure<T>();
        T& r = p.push(e, std::forward<A>(a)...);
        constructed(p, e);
        (void)r;
        return p.get(e);      // the element may have been moved to the front by an owning group
    }
    template <typename T, typename... A> decltype(auto) emplace_or_replace(entity e, A&&... a) {
        auto& p = assure<T>();
        if (p.contains(e)) { T& r = p.get(e); r = T(std::forward<A>(a)...); return (r); }
        return emplace<T>(e, std::forward<A>(a)...);
    }
    template <typename... T> bool any_of(entity e) { return (assure<T>().contains(e) || ...); }
    template <typename... T> bool all_of(entity e) { return (assure<T>().contains(e) && ...); }
    template <typename T> T& get(entity e) { return assure<T>().get(e); }
    template <typename T> T* try_get(entity e) { auto& p = assure<T>(); return p.contains(e) ? &p.get(e) : nullptr; }
    template <typename T> std::size_t remove(entity e) {
        auto& p = assure<T>();
        if (!p.contains(e)) return 0;
        destroying(p, e); p.erase_raw(e);
        return 1;
    }
    template <typename... T> void clear() {
        if constexpr (sizeof...(T) == 0) {
            for (auto& kv : pools) while (kv.second->size()) { entity e = kv.second->packed.back(); destroying(*kv.second, e); kv.second->erase_raw(e); }
            for (std::uint32_t i = 0; i < entities.size(); ++i) if (detail::idx(entities[i]) == i && valid(entities[i])) { entity e = entities[i]; entities[i] = compose(free_head, (version(e) + 1) & 0xFFFu); free_head = i; }
        } else {
            ([&] { auto& p = assure<T>(); while (p.size()) { entity e = p.packed.back(); destroying(p, e); p.erase_raw(e); } }(), ...);
        }
    }
    template <typename T> detail::pool<T>& storage() { return assure<T>(); }

    template <typename... T> view_t<T...> view() { return view_t<T...>{std::tuple<detail::pool<T>*...>{&assure<T>()...}}; }

    template <typename... O, typename... G, typename... X>
    group_t<std::tuple<O...>, std::tuple<G...>> group(get_t<G...> = {}, exclude_t<X...> = {}) {
        std::vector<std::type_index> key{std::type_index(typeid(O))..., std::type_index(typeid(void)), std::type_index(typeid(G))..., std::type_index(typeid(void)), std::type_index(typeid(X))...};
        detail::group_base* found = nullptr;
        for (auto& g : groups) if (g->key == key) found = g.get();
        if (!found) {
            auto g = std::make_unique<detail::group_base>();
            g->key = key;
            g->owned = {static_cast<detail::pool_base*>(&assure<O>())...};
            g->required = {static_cast<detail::pool_base*>(&assure<O>())..., static_cast<detail::pool_base*>(&assure<G>())...};
            g->excluded = {static_cast<detail::pool_base*>(&assure<X>())...};
            found = g.get();
            if constexpr (sizeof...(O) > 0) {
                // observers stay sorted least restrictive first, so that an entity entering several nested groups is moved
                // into the outer range before the inner one
                auto restrictiveness = [](detail::group_base* x) { return x->required.size() + x->excluded.size(); };
                for (auto* p : found->required) {
                    auto it = p->observers.begin();
                    while (it != p->observers.end() && restrictiveness(*it) <= restrictiveness(found)) ++it;
                    p->observers.insert(it, found);
                }
                // EnTT initialises a late group by walking the first owned pool in packed order
                detail::pool_base* first = found->owned[0];
                for (std::size_t i = 0; i < first->packed.size(); ++i) found->enter(first->packed[i]);
            }
            groups.push_back(std::move(g));
        }
        return group_t<std::tuple<O...>, std::tuple<G...>>{found, {&assure<O>()..., &assure<G>()...}};
    }

    context& ctx() { return vars; }

    // entity bookkeeping used by game_scene::forEachEntity / cloneTo (declared for completeness)
    template <typename F> void each(F f) { for (std::uint32_t i = 0; i < entities.size(); ++i) if (detail::idx(entities[i]) == i && valid(entities[i])) f(entities[i]); }
    const entity* data() const { return entities.data(); }
    std::size_t size() const { return entities.size(); }
    entity released() const { return compose(free_head, 0); }
    template <typename It> void assign(It first, It last, entity destroyed) { entities.assign(first, last); free_head = detail::idx(destroyed); }
    template <typename T, typename EIt, typename CIt> void insert(EIt first, EIt last, CIt from) { for (; first != last; ++first, ++from) emplace<T>(*first, *from); }

    template <typename T> entity owner_of(const T& c) { auto& p = assure<T>(); return p.packed[std::size_t(&c - p.data)]; }
};

template <typename T> entity to_entity(registry& r, const T& component) { return r.owner_of(component); }

} // namespace entt
