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
#include <unordered_map>
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

struct Ins {
    uint64_t key;   // src << 32 | dst
    uint64_t seq;   // insertion order
    double w;
};

}  // namespace el_detail

inline int read_edgelist(const char *path, bool weighted, bool directed, const char *delimiter, EdgeList &out) {
    using namespace el_detail;
    const size_t dl = delimiter ? strlen(delimiter) : 0;
    if (dl == 0) return EL_NEEDS_SLOW_READER;   // str.split("") raises in Python
    for (size_t i = 0; i < dl; i++)
        if ((unsigned char)delimiter[i] >= 0x80 || delimiter[i] == '\n' || delimiter[i] == '\r') return EL_NEEDS_SLOW_READER;

    FILE *f = fopen(path, "rb");
    if (!f) return EL_IO_ERROR;
    std::string buf;
    {
        char chunk[1 << 16];
        size_t got;
        while ((got = fread(chunk, 1, sizeof(chunk), f)) > 0) buf.append(chunk, got);
        const bool bad = ferror(f) != 0;
        fclose(f);
        if (bad) return EL_IO_ERROR;
    }
    // bytes whose treatment differs between this reader and Python's text layer / str.strip()
    for (size_t i = 0; i < buf.size(); i++) {
        const unsigned char c = (unsigned char)buf[i];
        if (c >= 0x80 || (c < 0x20 && c != '\t' && c != '\n' && c != '\r')) return EL_NEEDS_SLOW_READER;
        if (c == '\r' && (i + 1 >= buf.size() || buf[i + 1] != '\n')) return EL_NEEDS_SLOW_READER;   // lone CR = newline
    }

    std::unordered_map<Sv, uint32_t, SvHash> idmap;
    std::vector<Sv> ids;
    std::vector<Ins> ins;
    auto vertex = [&](Sv id) -> uint32_t {
        auto it = idmap.find(id);
        if (it != idmap.end()) return it->second;
        const uint32_t idx = (uint32_t)ids.size();
        idmap.emplace(id, idx);
        ids.push_back(id);
        return idx;
    };

    uint64_t seq = 0;
    size_t pos = 0;
    const size_t end = buf.size();
    while (pos < end) {
        size_t nl = pos;
        while (nl < end && buf[nl] != '\n') nl++;
        Sv line = strip(Sv{buf.data() + pos, nl - pos});
        pos = nl + 1;
        // split(delimiter) of the stripped line
        Sv terms[3];
        size_t n_terms = 0;
        size_t a = 0;
        for (;;) {
            size_t b = a;
            bool found = false;
            while (b + dl <= line.n) {
                if (memcmp(line.p + b, delimiter, dl) == 0) { found = true; break; }
                b++;
            }
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
        if (ids.size() + 2 >= 0xffffffffull) return EL_NEEDS_SLOW_READER;
        const uint32_t u = vertex(id1);
        const uint32_t v = vertex(id2);
        ins.push_back(Ins{((uint64_t)u << 32) | v, seq++, w});
        if (!directed) ins.push_back(Ins{((uint64_t)v << 32) | u, seq++, w});
    }

    std::sort(ins.begin(), ins.end(), [](const Ins &x, const Ins &y) {
        return x.key != y.key ? x.key < y.key : x.seq < y.seq;
    });
    const size_t n = ids.size();
    out.indptr.assign(n + 1, 0);
    out.indices.clear();
    out.data.clear();
    out.data64.clear();
    out.indices.reserve(ins.size());
    out.data.reserve(ins.size());
    out.data64.reserve(ins.size());
    for (size_t i = 0; i < ins.size();) {
        size_t j = i;
        while (j + 1 < ins.size() && ins[j + 1].key == ins[i].key) {
            j++;
            // an edge given twice with different weights triggers the reference's overwrite warning
            if (ins[j].w != ins[i].w) return EL_NEEDS_SLOW_READER;
        }
        out.indices.push_back((uint32_t)ins[j].key);
        out.data.push_back((float)ins[j].w);
        out.data64.push_back(ins[j].w);
        out.indptr[(size_t)(ins[i].key >> 32) + 1]++;
        i = j + 1;
    }
    if (out.indices.size() >= 0xffffffffull) return EL_NEEDS_SLOW_READER;
    for (size_t i = 0; i < n; i++) out.indptr[i + 1] += out.indptr[i];
    out.insertions = seq;
    out.id_off.assign(n + 1, 0);
    out.id_chars.clear();
    for (size_t i = 0; i < n; i++) {
        out.id_chars.append(ids[i].p, ids[i].n);
        out.id_off[i + 1] = out.id_chars.size();
    }
    return EL_OK;
}

}  // namespace pw
