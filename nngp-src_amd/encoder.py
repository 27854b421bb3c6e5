"""Query-line parsing and encoding (SURVEY.md 8a row a9 and "next" row N2).

Restates the reference's encoders without needing the table CSVs at run time -- only per-column
metadata (name, type, min/max, category count):

* single table: ``GeneralQuerySampler.parse_line / load_queries / transform_to_1d_array``
  (QuerySampler.py:157-221): line ``COL,upper,lower#COL,upper,lower@card``; numerical column i ->
  ``x[2i] = (upper-min)/(max-min)*1000``, ``x[2i+1] = (lower-min)/(max-min)*1000``; un-queried numerical
  columns default to (0, 1000); ``Y = log2(card)``.
* multi-join: ``Table`` / ``NNGPEncoder`` (neuroestimator/estimator/encoder.py:13-304): per-table
  predicate encodings concatenated in schema order followed by 3 slots per join triple, of which only
  the '=' slot is ever set (encoder.py:180-195); training line ``t1,t2@preds@preds@joins@card``, serving
  line the same without ``@card`` (encoder.py:207-250).
"""
from __future__ import annotations

import collections
import math
import os

import numpy as np

from .util import QueryInfo

Address = collections.namedtuple("Address", ["start", "end"])
JoinInfo = collections.namedtuple("JoinInfo", ["t1_id", "t2_id", "col_name", "col_type"])
ColumnSpec = collections.namedtuple("ColumnSpec", ["name", "kind", "lo", "hi", "num_categories"])

# UCI Covertype column ranges for forest.csv columns 0-9 (the reference takes min/max from the CSV,
# datasets.py:292-299 + QuerySampler.py:49-53; the CSV is not shipped, the published ranges are).
FOREST_COLUMNS = [ColumnSpec(n, "numerical", lo, hi, 0) for n, lo, hi in [
    ("A", 1859, 3858), ("B", 0, 360), ("C", 0, 66), ("D", 0, 1397), ("E", -173, 601),
    ("F", 0, 7117), ("G", 0, 254), ("H", 0, 254), ("I", 0, 254), ("J", 0, 7173)]]


def numerical(name, lo, hi):
    return ColumnSpec(name, "numerical", float(lo), float(hi), 0)


def categorical(name, num_categories):
    return ColumnSpec(name, "categorical", 0.0, 0.0, int(num_categories))


class TableEncoder(object):
    """Per-table predicate encoder (encoder.py:13-134 ``Table``; QuerySampler.py:16-57)."""

    def __init__(self, table_name, columns, chunk_size=64, zero_range_denominator=1e-6):
        self.table_name = table_name
        self.columns = list(columns)
        self.col_names = [c.name for c in self.columns]
        self.col_types = [c.kind for c in self.columns]
        self.num_cols = len(self.columns)
        self.chunk_size = chunk_size
        self.all_col_ranges = np.zeros((self.num_cols, 2))
        self.all_col_denominator = np.zeros((self.num_cols,))
        self.all_col_address = []
        self.table_feat_dim = 0
        for i, c in enumerate(self.columns):
            if c.kind == "categorical":
                encode_dim = math.ceil(float(c.num_categories) / chunk_size)
            else:
                self.all_col_ranges[i] = (c.lo, c.hi)
                den = c.hi - c.lo
                # encoder.py:56-57 guards a zero range with 1e-6; QuerySampler.py:215-218 does not
                self.all_col_denominator[i] = den if (den > 0 or zero_range_denominator is None) else zero_range_denominator
                encode_dim = 2
            self.all_col_address.append(Address(self.table_feat_dim, self.table_feat_dim + encode_dim))
            self.table_feat_dim += encode_dim

    @classmethod
    def from_dataframe(cls, df, col_types, table_name, chunk_size=64):
        """Column metadata the way the reference derives it from the CSV (encoder.py:29-59)."""
        df = df.fillna(-1)
        cols = []
        for name, kind in zip(df.columns, col_types):
            s = df[name]
            if kind == "categorical":
                cols.append(categorical(name, s.nunique()))
            else:
                cols.append(numerical(name, s.min(), s.max()))
        return cls(table_name, cols, chunk_size)

    def col_index(self, col_name):
        return self.col_names.index(col_name)

    def parse_predicates(self, pred_str):
        pred_list = []
        if not pred_str:
            return pred_list
        for predicate in pred_str.split("#"):
            items = predicate.split(",")
            col_idx = self.col_index(items[0].strip())
            if self.col_types[col_idx] == "categorical":
                pred_list.append((col_idx, [int(v.strip()) for v in items[1:]]))
            else:
                pred_list.append((col_idx, float(items[1].strip()), float(items[2].strip())))
        return pred_list

    def _factorized_encoding(self, col_idx, cat_set):
        assert self.col_types[col_idx] == "categorical", "Only categorical attribute supports factorized encoding"
        addr = self.all_col_address[col_idx]
        encode_dim = addr.end - addr.start
        bits = np.zeros(encode_dim * self.chunk_size, dtype=np.int64)
        for cat in cat_set:
            bits[int(cat)] = 1
        out = []
        for i in range(encode_dim):
            v = 0
            for b in bits[i * self.chunk_size:(i + 1) * self.chunk_size]:  # int(bitstring, 2): MSB first
                v = (v << 1) | int(b)
            out.append(v)
        return out

    def predicate_encoding(self, pred_list):
        x = np.zeros((self.table_feat_dim,), dtype=np.float64)
        for col_idx in range(self.num_cols):
            if self.col_types[col_idx] == "numerical":
                x[self.all_col_address[col_idx].start + 1] = 1000
        for pred in pred_list:
            col_idx = pred[0]
            addr = self.all_col_address[col_idx]
            if self.col_types[col_idx] == "categorical":
                x[addr.start:addr.end] = self._factorized_encoding(col_idx, pred[1])
            else:
                lo, den = self.all_col_ranges[col_idx][0], self.all_col_denominator[col_idx]
                x[addr.start] = (pred[1] - lo) / den * 1000
                x[addr.start + 1] = (pred[2] - lo) / den * 1000
        return x


class GeneralQueryEncoder(TableEncoder):
    """Single-table loader (QuerySampler.py:157-221)."""

    def __init__(self, columns=None, dataset="forest", chunk_size=64):
        super().__init__(dataset, FOREST_COLUMNS if columns is None else columns, chunk_size,
                         zero_range_denominator=None)
        self.total_feat_dim = self.table_feat_dim

    def parse_line(self, line):
        head, card = line.split("@")[0].strip(), int(line.split("@")[1].strip())
        return self.parse_predicates(head), card

    def load_queries(self, query_path, verbose=True):
        all_queries, all_cards, all_query_infos = [], [], []
        for sub_dir in sorted(os.listdir(query_path)):  # query_10.txt sorts before query_2.txt
            if verbose:
                print(sub_dir)
            with open(os.path.join(query_path, sub_dir), "r") as in_file:
                for line in in_file:
                    pred_list, card = self.parse_line(line)
                    all_queries.append(pred_list)
                    all_cards.append(card)
                    all_query_infos.append(QueryInfo(1, 0, len(pred_list), False, False))
        return all_queries, all_cards, all_query_infos

    def transform_to_1d_array(self, pred_list):
        return self.predicate_encoding(pred_list)

    def transform_to_arrays(self, all_queries, all_cards):
        X = np.array([self.transform_to_1d_array(p) for p in all_queries]).reshape(len(all_queries), self.total_feat_dim)
        Y = np.log2(np.reshape(np.array(all_cards), (len(all_queries), 1)))
        return X, Y


class NNGPEncoder(object):
    """Multi-join encoder (encoder.py:138-304)."""

    def __init__(self, tables, verbose=False):
        self.tables = list(tables)
        self.num_tables = len(self.tables)
        self.table_name_to_tid = {t.table_name: i for i, t in enumerate(self.tables)}
        self.all_join_infos = []
        for t1_id in range(self.num_tables - 1):
            t1 = self.tables[t1_id]
            for t2_id in range(t1_id + 1, self.num_tables):
                t2 = self.tables[t2_id]
                for ci, col_name in enumerate(t1.col_names):
                    if col_name in t2.col_names and t1.col_types[ci] == t2.col_types[t2.col_index(col_name)]:
                        self.all_join_infos.append(JoinInfo(t1_id, t2_id, col_name, t1.col_types[ci]))
        self.all_join_triples = [(j.t1_id, j.t2_id, j.col_name) for j in self.all_join_infos]
        self.join_ops_dict = {'>': 0, '<': 1, '=': 2}
        self.total_num_joins = len(self.all_join_triples)
        self.join_feat_dim = self.total_num_joins * len(self.join_ops_dict)
        self.feat_dim = sum(t.table_feat_dim for t in self.tables) + self.join_feat_dim
        if verbose:
            print("join feat dim = {}".format(self.join_feat_dim))

    def join_encoding(self, join_infos):
        join_x = np.zeros((self.join_feat_dim,), dtype=np.float64)
        for j in join_infos:
            triple = (j.t1_id, j.t2_id, j.col_name) if j.t1_id < j.t2_id else (j.t2_id, j.t1_id, j.col_name)
            idx = self.all_join_triples.index(triple)
            join_x[idx * len(self.join_ops_dict) + self.join_ops_dict['=']] = 1  # only '=' is ever encoded
        return join_x

    def transform_to_1d_array(self, table_ids, all_pred_list, join_infos):
        enc = []
        for t_id in range(self.num_tables):
            pred_list = all_pred_list[table_ids.index(t_id)] if t_id in table_ids else []
            enc.append(self.tables[t_id].predicate_encoding(pred_list))
        enc.append(self.join_encoding(join_infos))
        return np.hstack(enc)

    def _parse_terms(self, terms, table_ids, join_str):
        all_pred_list = [self.tables[t].parse_predicates(s.strip()) for t, s in zip(table_ids, terms[1:len(table_ids) + 1])]
        join_infos = []
        for join in ([] if not join_str else join_str.split('#')):
            t1_name, t2_name, col_name = [v.strip() for v in join.split(',')[:3]]
            t_id = self.table_name_to_tid[t1_name]
            col_type = self.tables[t_id].col_types[self.tables[t_id].col_index(col_name)]
            join_infos.append(JoinInfo(t_id, self.table_name_to_tid[t2_name], col_name, col_type))
        return all_pred_list, join_infos

    def parse_line(self, line):
        terms = line.strip().split('@')
        table_ids = [self.table_name_to_tid[n] for n in terms[0].strip().split(',')]
        assert len(table_ids) + 3 == len(terms), "Query Format Error!"
        all_pred_list, join_infos = self._parse_terms(terms, table_ids, terms[-2].strip())
        return table_ids, all_pred_list, join_infos, int(terms[-1].strip())

    def parse_line_without_card_then_encode(self, line):
        terms = line.strip().split('@')
        table_ids = [self.table_name_to_tid[n] for n in terms[0].strip().split(',')]
        assert len(table_ids) + 2 == len(terms), "Query Format Error!"
        all_pred_list, join_infos = self._parse_terms(terms, table_ids, terms[-1].strip())
        return self.transform_to_1d_array(table_ids, all_pred_list, join_infos)

    def _append(self, line, all_queries, all_cards, all_query_infos):
        table_ids, all_pred_list, join_infos, card = self.parse_line(line)
        all_queries.append((table_ids, all_pred_list, join_infos))
        all_cards.append(card)
        table_pairs = set((j.t1_id, j.t2_id) for j in join_infos)
        all_query_infos.append(QueryInfo(len(table_ids), len(join_infos), sum(len(p) for p in all_pred_list), True,
                                         len(table_pairs) < len(join_infos)))

    def load_queries(self, query_path, use_aux=False, q_error_threshold=100.0, coef_var_threshold=1.0):
        all_queries, all_cards, all_query_infos = [], [], []
        for sub_dir in sorted(os.listdir(query_path)):
            if sub_dir == 'join_query_aux.txt':
                if not use_aux:
                    continue
                with open(os.path.join(query_path, sub_dir), 'r') as in_file:
                    for line in in_file:
                        items = line.strip().split('@')
                        q_error, coef_var = float(items[-2]), float(items[-1])
                        if q_error < q_error_threshold and coef_var < coef_var_threshold:
                            continue  # well-predicted auxiliary queries are dropped (encoder.py:268-269)
                        self._append('@'.join(items[:-2]), all_queries, all_cards, all_query_infos)
                continue
            with open(os.path.join(query_path, sub_dir), "r") as in_file:
                for line in in_file:
                    self._append(line, all_queries, all_cards, all_query_infos)
        return all_queries, all_cards, all_query_infos

    def transform_to_arrays(self, all_queries, all_cards):
        X = np.array([self.transform_to_1d_array(*q) for q in all_queries]).reshape(len(all_queries), self.feat_dim)
        Y = np.log2(np.reshape(np.array(all_cards), (len(all_queries), 1)))
        return X, Y


class NativeEncoder(object):
    """The same encodings computed by the C++ encoder inside libnngp_hip.so: one call parses and
    encodes a whole batch of query lines, so the serving path is not bound by per-line Python."""

    def __init__(self, tables, chunk_size=64, single_table=False):
        import ctypes
        from . import _lib
        self._lib, self._ct = _lib, ctypes
        lines = []
        for t in tables:
            lines.append("table %s" % t.table_name)
            for c in t.columns:
                if c.kind == "categorical":
                    lines.append("cat %s %d" % (c.name, c.num_categories))
                else:
                    lines.append("num %s %r %r" % (c.name, float(c.lo), float(c.hi)))
        self.handle = ctypes.c_void_p()
        _lib.check(_lib.load().nngp_encoder_create(ctypes.byref(self.handle), "\n".join(lines).encode(), int(chunk_size),
                                                   1 if single_table else 0))
        self.feat_dim = int(_lib.load().nngp_encoder_dim(self.handle))

    @classmethod
    def from_encoder(cls, enc):
        """From a GeneralQueryEncoder (single table) or an NNGPEncoder (multi join)."""
        if isinstance(enc, NNGPEncoder):
            return cls(enc.tables, enc.tables[0].chunk_size, single_table=False)
        return cls([enc], enc.chunk_size, single_table=True)

    def encode_lines(self, lines, with_card=False):
        """lines: iterable of str -> X [n, d] float64 (and cards [n] when with_card)."""
        ct = self._ct
        text = "\n".join(l.strip("\n") for l in lines).encode()
        cap = text.count(b"\n") + 1
        x = np.empty((cap, self.feat_dim), dtype=np.float64)
        cards = np.empty((cap,), dtype=np.float64)
        n = ct.c_int64(0)
        self._lib.check(self._lib.load().nngp_encoder_encode(self.handle, text, len(text), int(bool(with_card)),
                                                             x.ctypes.data_as(ct.c_void_p), cards.ctypes.data_as(ct.c_void_p),
                                                             cap, ct.byref(n)))
        x = x[: n.value]
        return (x, cards[: n.value]) if with_card else x

    def close(self):
        if getattr(self, "handle", None):
            self._lib.load().nngp_encoder_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
