#!/usr/bin/env python3
"""Child program of the GPU tier (no GPU context): the oracle's answers to many queries on several cores.

argv: terms file, dtmap file, jobs.json (a list of [query, algo, limit, fuzzymatch]), number of workers.
Loads the oracle index ONCE, forks the workers (they share its pages), prints one JSON line: a list of
result lists [[doc, score], ...] in job order (None where the oracle raised).  Test infrastructure, like
everything under oracle/: the product never runs it."""
import json
import multiprocessing as mp
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib as O

IDX = None


def one(job):
    q, algo, limit, fz = job
    try:
        return [[int(d), float(s)] for d, s in IDX.search(q, algo=algo, limit=limit, fuzzymatch=fz)]
    except O.SearchError:
        return None


def main():
    global IDX
    terms, dtmap, jobs_path, n = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    jobs = json.load(open(jobs_path))
    IDX = O.Index(terms, dtmap)
    with mp.get_context("fork").Pool(n) as pool:
        out = pool.map(one, jobs, chunksize=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
