"""Schema loaders for the serving path (reference neuroestimator/estimator/util.py:159-195).

The reference reads the TPC-DS / IMDB CSVs to derive per-column min/max and category counts.  Those
benchmark tables are data, not part of the hot path, and are not shipped; this module only needs the
column metadata, which it derives from CSVs when ``data_path`` holds them.
"""
from __future__ import annotations

import os

from .encoder import NNGPEncoder, TableEncoder

# schema name -> [(table name, csv file, [column names], [column kinds])]; users register their own.
SCHEMAS = {}


def register_schema(name, tables):
    SCHEMAS[name] = tables


def load_training_schema_data(schema_name, data_path, query_path, chunk_size=64, use_aux=False,
                              q_error_threshold=100.0, coef_var_threshold=1.0):
    if schema_name not in SCHEMAS:
        raise NotImplementedError(
            "schema %r is not registered: register_schema(name, [(table, csv, columns, kinds), ...]) or pass "
            "encoder= to Estimator (the reference's tpcds/imdb column lists need the benchmark CSVs)" % schema_name)
    import pandas as pd
    tables = []
    for table_name, csv_file, columns, kinds in SCHEMAS[schema_name]:
        path = os.path.join(data_path, csv_file)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        df = pd.read_csv(path, usecols=columns)
        tables.append(TableEncoder.from_dataframe(df[columns], kinds, table_name, chunk_size))
    encoder = NNGPEncoder(tables)
    queries, cards, _ = encoder.load_queries(query_path, use_aux, q_error_threshold, coef_var_threshold)
    X, Y = encoder.transform_to_arrays(queries, cards)
    return X, Y, encoder
