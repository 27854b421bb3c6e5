// Native query-line parser + encoder (host C++).
//
// Restates, without pandas, the per-line work the reference does in Python on the serving path
// (neuroestimator/estimator/encoder.py:59-97,187-250 `Table.parse_predicates / predicate_encoding`,
// `NNGPEncoder.join_encoding / transform_to_1d_array / parse_line*`) and the single-table loader
// (QuerySampler.py:157-221).  Results are bit-identical to those encoders (same float64 operations in the
// same order; golden vectors produced by the reference's own encoder.py: tests/golden/encoder_ref.json).
//
// The schema arrives as text (one directive per line), which keeps the C ABI free of nested structs:
//     table <name>
//     num <column> <min> <max>
//     cat <column> <num_categories>
// mode 0 = multi-join lines  "t1,t2@preds_t1@preds_t2@t1,t2,col#...[@card]"   (zero ranges use 1e-6, encoder.py:56-57)
// mode 1 = single-table lines "COL,upper,lower#COL,upper,lower@card"          (QuerySampler.py: no zero-range guard)
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/nngp_hip.h"

namespace nngp {
void set_error(const char* fmt, ...);
}

namespace {

struct Column {
    std::string name;
    bool categorical = false;
    int64_t num_categories = 0;
    double lo = 0.0, denom = 1.0;
    int start = 0, width = 0;  // slots inside the table's block
};

struct Table {
    std::string name;
    std::vector<Column> cols;
    int dim = 0, offset = 0;  // offset of the table's block in the feature vector
    int find(const std::string& c) const {
        for (size_t i = 0; i < cols.size(); ++i)
            if (cols[i].name == c) return (int)i;
        return -1;
    }
};

struct JoinTriple {
    int t1, t2;
    std::string col;
};

std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r' || s[a] == '\n')) ++a;
    while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r' || s[b - 1] == '\n')) --b;
    return s.substr(a, b - a);
}

std::vector<std::string> split(const std::string& s, char sep) {
    std::vector<std::string> out;
    size_t start = 0;
    for (;;) {
        const size_t p = s.find(sep, start);
        if (p == std::string::npos) {
            out.push_back(s.substr(start));
            return out;
        }
        out.push_back(s.substr(start, p - start));
        start = p + 1;
    }
}

}  // namespace

struct nngp_encoder {
    std::vector<Table> tables;
    std::vector<JoinTriple> joins;
    int chunk = 64, mode = 0, dim = 0, join_offset = 0;

    int table_id(const std::string& n) const {
        for (size_t i = 0; i < tables.size(); ++i)
            if (tables[i].name == n) return (int)i;
        return -1;
    }

    // encoder.py:76-97 predicate_encoding into x[table.offset ...]; returns false on a format error
    bool encode_predicates(const Table& t, const std::string& pred_str, double* x) const {
        double* xt = x + t.offset;
        for (const Column& c : t.cols)
            if (!c.categorical) {
                xt[c.start] = 0.0;
                xt[c.start + 1] = 1000.0;
            } else {
                for (int k = 0; k < c.width; ++k) xt[c.start + k] = 0.0;
            }
        if (pred_str.empty()) return true;
        for (const std::string& pred : split(pred_str, '#')) {
            const std::vector<std::string> items = split(pred, ',');
            const int ci = t.find(trim(items[0]));
            if (ci < 0) {
                nngp::set_error("encoder: unknown column '%s' in table '%s'", trim(items[0]).c_str(), t.name.c_str());
                return false;
            }
            const Column& c = t.cols[ci];
            if (c.categorical) {
                // factorised encoding: one bit per category, read back as chunk-size-bit big-endian integers
                std::vector<uint64_t> words(c.width, 0);
                for (size_t k = 1; k < items.size(); ++k) {
                    const long long cat = strtoll(trim(items[k]).c_str(), nullptr, 10);
                    if (cat < 0 || cat >= (long long)c.width * chunk) {
                        nngp::set_error("encoder: category %lld out of range for column '%s'", cat, c.name.c_str());
                        return false;
                    }
                    words[cat / chunk] |= (uint64_t)1 << (chunk - 1 - (int)(cat % chunk));
                }
                for (int k = 0; k < c.width; ++k) xt[c.start + k] = (double)words[k];
            } else {
                if (items.size() < 3) {
                    nngp::set_error("encoder: numerical predicate needs 'col,upper,lower': '%s'", pred.c_str());
                    return false;
                }
                const double upper = strtod(trim(items[1]).c_str(), nullptr), lower = strtod(trim(items[2]).c_str(), nullptr);
                xt[c.start] = (upper - c.lo) / c.denom * 1000;
                xt[c.start + 1] = (lower - c.lo) / c.denom * 1000;
            }
        }
        return true;
    }

    bool encode_line(const std::string& raw, bool with_card, double* x, double* card) const {
        const std::string line = trim(raw);
        std::vector<std::string> terms = split(line, '@');
        if (mode == 1) {  // QuerySampler.parse_line: "preds@card"
            if (with_card) {
                if (terms.size() < 2) { nngp::set_error("encoder: missing '@card': '%s'", line.c_str()); return false; }
                *card = (double)strtoll(trim(terms[1]).c_str(), nullptr, 10);
            }
            return encode_predicates(tables[0], trim(terms[0]), x);
        }
        std::vector<int> tids;
        for (const std::string& n : split(trim(terms[0]), ',')) {
            const int id = table_id(trim(n));
            if (id < 0) { nngp::set_error("encoder: unknown table '%s'", trim(n).c_str()); return false; }
            tids.push_back(id);
        }
        if (tids.size() + (with_card ? 3 : 2) != terms.size()) {
            nngp::set_error("Query Format Error!");  // encoder.py:212,234
            return false;
        }
        for (size_t t = 0; t < tables.size(); ++t) {  // absent tables get their default encoding (encoder.py:197-205)
            std::string preds;
            for (size_t k = 0; k < tids.size(); ++k)
                if (tids[k] == (int)t) { preds = trim(terms[1 + k]); break; }
            if (!encode_predicates(tables[t], preds, x)) return false;
        }
        for (size_t k = 0; k < joins.size() * 3; ++k) x[join_offset + k] = 0.0;
        const std::string join_str = trim(terms[with_card ? terms.size() - 2 : terms.size() - 1]);
        if (!join_str.empty())
            for (const std::string& j : split(join_str, '#')) {
                const std::vector<std::string> it = split(j, ',');
                if (it.size() < 3) { nngp::set_error("encoder: bad join '%s'", j.c_str()); return false; }
                int a = table_id(trim(it[0])), b = table_id(trim(it[1]));
                const std::string col = trim(it[2]);
                if (a < 0 || b < 0) { nngp::set_error("encoder: unknown table in join '%s'", j.c_str()); return false; }
                if (a > b) { const int t = a; a = b; b = t; }
                int idx = -1;
                for (size_t q = 0; q < joins.size(); ++q)
                    if (joins[q].t1 == a && joins[q].t2 == b && joins[q].col == col) { idx = (int)q; break; }
                if (idx < 0) { nngp::set_error("encoder: '%s' is not a join of the schema", j.c_str()); return false; }
                x[join_offset + idx * 3 + 2] = 1.0;  // only '=' is ever encoded (encoder.py:187-195)
            }
        if (with_card) *card = (double)strtoll(trim(terms.back()).c_str(), nullptr, 10);
        return true;
    }
};

extern "C" {

int nngp_encoder_create(nngp_encoder** out, const char* schema_text, int32_t chunk_size, int32_t mode) {
    if (!out || !schema_text || chunk_size < 1 || chunk_size > 64 || (mode != 0 && mode != 1)) {
        nngp::set_error("encoder_create: bad arguments (chunk_size must be in [1, 64], mode 0 or 1)");
        return -2;
    }
    nngp_encoder* e = new nngp_encoder();
    e->chunk = chunk_size;
    e->mode = mode;
    for (const std::string& raw : split(schema_text, '\n')) {
        const std::string line = trim(raw);
        if (line.empty()) continue;
        std::vector<std::string> w;
        for (const std::string& t : split(line, ' '))
            if (!t.empty()) w.push_back(t);
        if (w[0] == "table" && w.size() == 2) {
            Table t;
            t.name = w[1];
            e->tables.push_back(t);
        } else if (w[0] == "num" && w.size() == 4 && !e->tables.empty()) {
            Column c;
            c.name = w[1];
            c.lo = strtod(w[2].c_str(), nullptr);
            const double den = strtod(w[3].c_str(), nullptr) - c.lo;
            c.denom = (mode == 0 && !(den > 0)) ? 1e-6 : den;
            c.width = 2;
            e->tables.back().cols.push_back(c);
        } else if (w[0] == "cat" && w.size() == 3 && !e->tables.empty()) {
            Column c;
            c.name = w[1];
            c.categorical = true;
            c.num_categories = strtoll(w[2].c_str(), nullptr, 10);
            c.width = (int)((c.num_categories + chunk_size - 1) / chunk_size);
            e->tables.back().cols.push_back(c);
        } else {
            nngp::set_error("encoder_create: bad schema line '%s'", line.c_str());
            delete e;
            return -2;
        }
    }
    if (e->tables.empty() || (mode == 1 && e->tables.size() != 1)) {
        nngp::set_error("encoder_create: schema needs at least one table (exactly one in single-table mode)");
        delete e;
        return -2;
    }
    int off = 0;
    for (Table& t : e->tables) {
        t.offset = off;
        for (Column& c : t.cols) {
            c.start = t.dim;
            t.dim += c.width;
        }
        off += t.dim;
    }
    e->join_offset = off;
    if (mode == 0) {  // join triples: shared column names of equal type, table pairs in schema order (encoder.py:150-160)
        for (size_t a = 0; a + 1 < e->tables.size(); ++a)
            for (size_t b = a + 1; b < e->tables.size(); ++b)
                for (const Column& c : e->tables[a].cols) {
                    const int j = e->tables[b].find(c.name);
                    if (j >= 0 && e->tables[b].cols[j].categorical == c.categorical) e->joins.push_back({(int)a, (int)b, c.name});
                }
        off += (int)e->joins.size() * 3;
    }
    e->dim = off;
    *out = e;
    return 0;
}

int nngp_encoder_destroy(nngp_encoder* e) {
    delete e;
    return 0;
}

int32_t nngp_encoder_dim(const nngp_encoder* e) { return e ? e->dim : -1; }

/* Encodes '\n'-separated query lines.  x_out: HOST [max_lines, dim] float64; card_out: HOST [max_lines] (true
 * cardinalities, only when with_card != 0; may be NULL otherwise).  n_lines_out receives the number of lines encoded. */
int nngp_encoder_encode(const nngp_encoder* e, const char* text, int64_t text_len, int32_t with_card, double* x_out,
                        double* card_out, int64_t max_lines, int64_t* n_lines_out) {
    if (!e || !text || !x_out || !n_lines_out || (with_card && !card_out)) {
        nngp::set_error("encoder_encode: NULL argument");
        return -2;
    }
    int64_t n = 0, pos = 0;
    while (pos < text_len) {
        int64_t end = pos;
        while (end < text_len && text[end] != '\n') ++end;
        std::string line(text + pos, (size_t)(end - pos));
        pos = end + 1;
        if (trim(line).empty()) continue;
        if (n >= max_lines) {
            nngp::set_error("encoder_encode: more than %lld lines", (long long)max_lines);
            return -2;
        }
        double card = 0.0;
        if (!e->encode_line(line, with_card != 0, x_out + n * e->dim, &card)) return -3;
        if (with_card) card_out[n] = card;
        ++n;
    }
    *n_lines_out = n;
    return 0;
}

}  // extern "C"
