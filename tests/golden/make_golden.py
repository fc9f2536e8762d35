#!/usr/bin/env python3
"""Extracts the golden on-disk byte vectors held by the reference's own tests
(src/tests/t_index_terms.c:23-37, src/tests/t_index_dtmap.c:25-41) into
tests/golden/reference_vectors.json.  Run in the build container only (needs
/root/reference); the JSON it updates is committed and travels.

Only DATA (the expected byte arrays) is extracted, no source text.
"""
import json
import os
import re

REF = "/root/reference/src/tests"
HERE = os.path.dirname(os.path.abspath(__file__))


def byte_array(path, name):
    src = open(path).read()
    m = re.search(r"%s\[\]\s*=\s*\{(.*?)\};" % name, src, re.S)
    body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
    body = re.sub(r"//[^\n]*", "", body)
    return bytes(int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]{2}", body))


def raw_scoring_docs(path):
    """t_scoring.c:16-163: the RAW document texts and queries of the seven cases
    (data the reference's test holds), in case order."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    docs = {}
    for m in re.finditer(r"test_doc_t\s+(docs_\d+)\[\]\s*=\s*\{(.*?)\n\};", src, re.S):
        items = {}
        for e in re.finditer(r"\{\s*(\d+)\s*,((?:\s*\"[^\"]*\")+)\s*\}", m.group(2)):
            items[e.group(1)] = "".join(re.findall(r"\"([^\"]*)\"", e.group(2)))
        docs[m.group(1)] = items
    cases = []
    for m in re.finditer(r"test_search_case_t\s+test_case_\d+\s*=\s*\{(.*?)\n\};", src, re.S):
        body = m.group(1)
        cases.append({"docs": docs[re.search(r"\.docs\s*=\s*(docs_\d+)", body).group(1)],
                      "query": re.search(r"\.query\s*=\s*\"([^\"]*)\"", body).group(1)})
    return cases


def main():
    p = os.path.join(HERE, "reference_vectors.json")
    d = json.load(open(p))
    raw = raw_scoring_docs(os.path.join(REF, "t_scoring.c"))
    assert len(raw) == len(d["scoring"]["cases"])
    for r, c in zip(raw, d["scoring"]["cases"]):
        r["scores"] = c["scores"]
    d["scoring_raw"] = {
        "source": "src/tests/t_scoring.c:16-163: the RAW document texts and queries (the reference indexes and "
                  "queries them through its default filters: normalizer, stopwords -- no stop-word file in the "
                  "test's basedir --, stemmer); scores = [TF-IDF, BM25], tolerance 1e-4 (src/tests/helpers.c:215)",
        "tolerance": d["scoring"]["tolerance"], "cases": raw}
    d["terms_db"]["hex"] = byte_array(
        os.path.join(REF, "t_index_terms.c"), "terms_db_exp").hex()
    d["dtmap_db"]["hex"] = byte_array(
        os.path.join(REF, "t_index_dtmap.c"), "dtmap_db_exp").hex()
    json.dump(d, open(p, "w"), indent=1, ensure_ascii=False)
    print("terms_db", len(d["terms_db"]["hex"]) // 2, "bytes;",
          "dtmap_db", len(d["dtmap_db"]["hex"]) // 2, "bytes")


if __name__ == "__main__":
    main()
