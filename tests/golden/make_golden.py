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


def main():
    p = os.path.join(HERE, "reference_vectors.json")
    d = json.load(open(p))
    d["terms_db"]["hex"] = byte_array(
        os.path.join(REF, "t_index_terms.c"), "terms_db_exp").hex()
    d["dtmap_db"]["hex"] = byte_array(
        os.path.join(REF, "t_index_dtmap.c"), "dtmap_db_exp").hex()
    json.dump(d, open(p, "w"), indent=1, ensure_ascii=False)
    print("terms_db", len(d["terms_db"]["hex"]) // 2, "bytes;",
          "dtmap_db", len(d["dtmap_db"]["hex"]) // 2, "bytes")


if __name__ == "__main__":
    main()
