// edgelist.hpp -- host-side edge-list ingestion (no GPU involved): text file -> sorted CSR.
//
// Replaces the reference's dict-of-dicts reader (AdjlstGraph.read / add_edge / to_csr,
// src/pecanpy/graph.py:270-341) for WELL-FORMED files.  The reader is optimistic: it implements the
// reference's semantics for the common case and reports "needs the slow reader" for every input
// whose handling involves Python-specific behaviour (warnings, exotic float literals, malformed lines,
// non-ASCII / exotic whitespace) -- the Python side then falls back to its statement-by-statement reader,
// so the observable behaviour is the reference's in all cases.
//
//   - a line is `line.strip().split(delimiter)`; ids are terms[0].strip(), terms[1].strip()
//   - vertices are numbered in order of first appearance (id1 before id2)                   graph.py:218-238
//   - unweighted: weight 1.0, extra columns ignored; weighted: exactly three columns        graph.py:160-176
//   - undirected: the reverse edge is inserted with the same weight                         graph.py:266-268
//   - a repeated edge keeps the LAST weight; num_edges counts insertions, not distinct edges graph.py:240-243
//   - to_csr: rows in vertex order, neighbours ascending, float32 weights                   graph.py:323-341
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

namespace pw {

struct EdgeList {
    std::vector<uint32_t> indptr, indices;
    std::vector<float> data;
    std::vector<double> data64;     // the weights as parsed (DenseGraph keeps float64)
    std::vector<uint64_t> id_off;   // [n + 1] offsets into id_chars
    std::string id_chars;           // concatenated vertex ids (no separators)
    uint64_t insertions = 0;        // the reference's num_edges
};

enum EdgeListStatus { EL_OK = 0, EL_NEEDS_SLOW_READER = 1, EL_IO_ERROR = 2 };

namespace el_detail {

struct Sv {   // string view over the file buffer
    const char *p;
    size_t n;
    bool operator==(const Sv &o) const { return n == o.n && memcmp(p, o.p, n) == 0; }
};
struct SvHash {
    size_t operator()(const Sv &s) const {   // FNV-1a, 64 bit
        uint64_t h = 1469598103934665603ull;
        for (size_t i = 0; i < s.n; i++) h = (h ^ (unsigned char)s.p[i]) * 1099511628211ull;
        return (size_t)h;
    }
};

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }
inline Sv strip(Sv s) {
    while (s.n && is_space(s.p[0])) { s.p++; s.n--; }
    while (s.n && is_space(s.p[s.n - 1])) s.n--;
    return s;
}

// plain decimal literals only: [+-]digits[.digits][e[+-]digits]; everything else (inf, nan, 1_000,
// hex, empty) goes to the slow reader, which applies Python's float()
inline bool parse_weight(Sv t, double &w) {
    t = strip(t);
    if (t.n == 0 || t.n > 63) return false;
    size_t i = 0;
    if (t.p[i] == '+' || t.p[i] == '-') i++;
    size_t digits = 0;
    while (i < t.n && t.p[i] >= '0' && t.p[i] <= '9') { i++; digits++; }
    if (i < t.n && t.p[i] == '.') {
        i++;
        while (i < t.n && t.p[i] >= '0' && t.p[i] <= '9') { i++; digits++; }
    }
    if (digits == 0) return false;
    if (i < t.n && (t.p[i] == 'e' || t.p[i] == 'E')) {
        i++;
        if (i < t.n && (t.p[i] == '+' || t.p[i] == '-')) i++;
        size_t ed = 0;
        while (i < t.n && t.p[i] >= '0' && t.p[i] <= '9') { i++; ed++; }
        if (ed == 0) return false;
    }
    if (i != t.n) return false;
    char buf[64];
    memcpy(buf, t.p, t.n);
    buf[t.n] = 0;
    char *end = nullptr;
    w = strtod(buf, &end);   // correctly rounded, like Python's float()
    return end == buf + t.n;
}

}  // namespace el_detail

// Vertex ids -> indices in order of first appearance.  Canonical decimal ids ("0", "17", no sign, no leading
// zero, below 2^26) -- what most edge lists contain -- go through a direct value table; everything else
// through an open-addressing hash table (slot = entry + 1, 0 = empty).  Two ids are equal iff their strings
// are equal, and a canonical decimal string is determined by its value, so the two tables never overlap.
struct IdTable {
    std::vector<el_detail::Sv> ids;          // all ids, first-appearance order
    std::vector<uint32_t> by_value;          // canonical decimal value -> index + 1
    std::vector<uint64_t> hashes;            // hashed entries: hash, index into `ids`
    std::vector<uint32_t> entry_id;
    std::vector<uint32_t> slots;
    uint64_t mask = 0;
    static constexpr uint32_t VALUE_CAP = 1u << 26;
    IdTable() { slots.assign(1u << 12, 0u); mask = slots.size() - 1; }
    void grow() {
        std::vector<uint32_t> bigger(slots.size() * 2, 0u);
        const uint64_t m = bigger.size() - 1;
        for (uint32_t i = 0; i < hashes.size(); i++) {
            uint64_t p = hashes[i] & m;
            while (bigger[p]) p = (p + 1) & m;
            bigger[p] = i + 1;
        }
        slots.swap(bigger);
        mask = m;
    }
    uint32_t add(el_detail::Sv id) {
        ids.push_back(id);
        return (uint32_t)ids.size() - 1;
    }
    uint32_t get(el_detail::Sv id) {
        if (id.n >= 1 && id.n <= 8 && (id.n == 1 || id.p[0] != '0')) {
            uint32_t v = 0;
            size_t i = 0;
            for (; i < id.n; i++) {
                const unsigned c = (unsigned)id.p[i] - '0';
                if (c > 9) break;
                v = v * 10 + c;
            }
            if (i == id.n && v < VALUE_CAP) {
                if (v >= by_value.size()) by_value.resize(std::max<size_t>((size_t)v + 1, by_value.size() * 2), 0u);
                if (by_value[v] == 0) by_value[v] = add(id) + 1;
                return by_value[v] - 1;
            }
        }
        const uint64_t h = el_detail::SvHash()(id);
        uint64_t p = h & mask;
        while (slots[p]) {
            const uint32_t e = slots[p] - 1;
            if (hashes[e] == h && ids[entry_id[e]] == id) return entry_id[e];
            p = (p + 1) & mask;
        }
        const uint32_t idx = add(id);
        hashes.push_back(h);
        entry_id.push_back(idx);
        slots[p] = (uint32_t)hashes.size();
        if (hashes.size() * 2 > slots.size()) grow();
        return idx;
    }
};

inline int read_edgelist(const char *path, bool weighted, bool directed, const char *delimiter, EdgeList &out) {
    using namespace el_detail;
    const size_t dl = delimiter ? strlen(delimiter) : 0;
    if (dl == 0) return EL_NEEDS_SLOW_READER;   // str.split("") raises in Python
    for (size_t i = 0; i < dl; i++)
        if ((unsigned char)delimiter[i] >= 0x80 || delimiter[i] == '\n' || delimiter[i] == '\r') return EL_NEEDS_SLOW_READER;
    const char d0 = delimiter[0];

    FILE *f = fopen(path, "rb");
    if (!f) return EL_IO_ERROR;
    std::string buf;
    {
        if (fseek(f, 0, SEEK_END) == 0) {
            const long sz = ftell(f);
            if (sz > 0) buf.reserve((size_t)sz + 1);
            rewind(f);
        }
        char chunk[1 << 16];
        size_t got;
        while ((got = fread(chunk, 1, sizeof(chunk), f)) > 0) buf.append(chunk, got);
        const bool bad = ferror(f) != 0;
        fclose(f);
        if (bad) return EL_IO_ERROR;
    }

    IdTable table;
    struct Raw { uint32_t u, v; double w; };
    std::vector<Raw> raw;          // one entry per line, file order
    raw.reserve(buf.size() / 8 + 16);
    Sv last1{nullptr, 0}, last2{nullptr, 0};
    uint32_t last_u = 0, last_v = 0;

    size_t pos = 0;
    const size_t end = buf.size();
    const char *base = buf.data();
    while (pos < end) {
        // one pass over the line: find its end and reject the bytes whose treatment differs between this
        // reader and Python's text layer / str.strip()
        size_t nl = pos;
        for (; nl < end; nl++) {
            const unsigned char c = (unsigned char)base[nl];
            if (c == '\n') break;
            if (c >= 0x80 || (c < 0x20 && c != '\t' && c != '\r')) return EL_NEEDS_SLOW_READER;
            if (c == '\r' && (nl + 1 >= end || base[nl + 1] != '\n')) return EL_NEEDS_SLOW_READER;   // lone CR = newline
        }
        Sv line = strip(Sv{base + pos, nl - pos});
        pos = nl + 1;
        // split(delimiter) of the stripped line
        Sv terms[3];
        size_t n_terms = 0;
        size_t a = 0;
        for (;;) {
            size_t b = a;
            bool found = false;
            for (; b + dl <= line.n; b++)
                if (line.p[b] == d0 && (dl == 1 || memcmp(line.p + b, delimiter, dl) == 0)) { found = true; break; }
            if (!found) b = line.n;
            if (n_terms < 3) terms[n_terms] = Sv{line.p + a, b - a};
            n_terms++;
            if (!found) break;
            a = b + dl;
        }
        if (n_terms < 2) return EL_NEEDS_SLOW_READER;              // IndexError in the reference
        if (weighted && n_terms != 3) return EL_NEEDS_SLOW_READER;  // ValueError in the reference
        double w = 1.0;
        if (weighted) {
            if (!parse_weight(terms[2], w)) return EL_NEEDS_SLOW_READER;
            if (!(w > 0.0)) return EL_NEEDS_SLOW_READER;           // "Non-positive edge ignored" warning
        }
        const Sv id1 = strip(terms[0]), id2 = strip(terms[1]);
        if (table.ids.size() + 2 >= 0xffffffffull) return EL_NEEDS_SLOW_READER;
        // edge lists are usually grouped by source: remember the previous line's ids
        const uint32_t u = (last1.p && last1 == id1) ? last_u : table.get(id1);
        const uint32_t v = (last2.p && last2 == id2) ? last_v : ((id2 == id1) ? u : table.get(id2));
        last1 = id1; last_u = u;
        last2 = id2; last_v = v;
        raw.push_back(Raw{u, v, w});
    }

    // bucket the insertions by source vertex, in insertion order (counting sort = stable)
    const size_t n = table.ids.size();
    const uint64_t n_ins = (uint64_t)raw.size() * (directed ? 1u : 2u);
    if (n_ins >= 0xffffffffull) return EL_NEEDS_SLOW_READER;
    std::vector<uint64_t> start(n + 1, 0);
    for (const Raw &e : raw) {
        start[e.u + 1]++;
        if (!directed) start[e.v + 1]++;
    }
    for (size_t i = 0; i < n; i++) start[i + 1] += start[i];
    struct Half { uint32_t dst; uint32_t seq; double w; };   // seq < 2^32 checked above
    std::vector<Half> half(n_ins);
    {
        std::vector<uint64_t> fill(start.begin(), start.end() - 1);
        uint32_t seq = 0;
        for (const Raw &e : raw) {
            half[fill[e.u]++] = Half{e.v, seq++, e.w};
            if (!directed) half[fill[e.v]++] = Half{e.u, seq++, e.w};
        }
    }
    std::vector<Raw>().swap(raw);

    out.indptr.assign(n + 1, 0);
    out.indices.clear();
    out.data.clear();
    out.data64.clear();
    out.indices.reserve(n_ins);
    out.data.reserve(n_ins);
    out.data64.reserve(n_ins);
    for (size_t r = 0; r < n; r++) {
        Half *b = half.data() + start[r], *e = half.data() + start[r + 1];
        bool sorted = true;
        for (Half *p = b; p + 1 < e; p++)
            if (p[1].dst <= p[0].dst) { sorted = false; break; }
        if (!sorted) std::sort(b, e, [](const Half &x, const Half &y) { return x.dst != y.dst ? x.dst < y.dst : x.seq < y.seq; });
        for (Half *p = b; p < e;) {
            Half *q = p;
            while (q + 1 < e && q[1].dst == p->dst) {
                q++;
                // an edge given twice with different weights triggers the reference's overwrite warning
                if (q->w != p->w) return EL_NEEDS_SLOW_READER;
            }
            out.indices.push_back(q->dst);        // the last insertion wins
            out.data.push_back((float)q->w);
            out.data64.push_back(q->w);
            p = q + 1;
        }
        out.indptr[r + 1] = (uint32_t)out.indices.size();
    }
    out.insertions = n_ins;
    out.id_off.assign(n + 1, 0);
    out.id_chars.clear();
    for (size_t i = 0; i < n; i++) {
        out.id_chars.append(table.ids[i].p, table.ids[i].n);
        out.id_off[i + 1] = out.id_chars.size();
    }
    return EL_OK;
}

}  // namespace pw
