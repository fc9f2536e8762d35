"""Writer for the reference's on-disk index format (test infrastructure).

Produces valid `nxsterms` / `nxsdtmap` files from pre-tokenised documents, the
way `nxs_index_add()` would (reference src/index/storage.h:13-134,
terms.c:155-318, dtmap.c:149-355): term ids are 1-based first-seen order, doc
blocks carry (term_id, count) pairs sorted by term id, all integers
big-endian, files sized in 32 KiB steps (index.h:24).
"""
import os
import struct

IDX_SIZE_STEP = 32 * 1024


def _pad32k(b: bytes) -> bytes:
    n = (len(b) + IDX_SIZE_STEP - 1) // IDX_SIZE_STEP * IDX_SIZE_STEP
    return b + b"\0" * (max(n, IDX_SIZE_STEP) - len(b))


def terms_image(terms, totals, pad=True) -> bytes:
    """terms: list[bytes] in term-id order; totals: list[int]."""
    body = b""
    for t, tot in zip(terms, totals):
        blk = struct.pack(">H", len(t)) + t + b"\0"
        blk += b"\0" * (-len(blk) % 8)
        blk += struct.pack(">Q", tot)
        body += blk
    hdr = b"NXS_T" + bytes([1, 0, 0]) + struct.pack(">II", len(body), 0)
    img = hdr + body
    return _pad32k(img) if pad else img


def dtmap_image(blocks, token_count, doc_count, pad=True) -> bytes:
    """blocks: list of (doc_id, doc_len, [(term_id, count), ...]);
    a tombstone is (doc_id, 0, [])."""
    body = b""
    for doc_id, doc_len, pairs in blocks:
        body += struct.pack(">QII", doc_id, doc_len, len(pairs))
        for tid, cnt in pairs:
            body += struct.pack(">II", tid, cnt)
    hdr = (b"NXS_D" + bytes([1, 0, 0]) +
           struct.pack(">QQII", len(body), token_count, doc_count, 0))
    img = hdr + body
    return _pad32k(img) if pad else img


def build_images(docs, removed=()):
    """docs: iterable of (doc_id, [token bytes...]).  `removed`: doc ids
    deleted afterwards via the reference's protocol (block id zeroed +
    tombstone appended, counters decremented: dtmap.c:546-658)."""
    term_ids = {}
    terms, totals = [], []
    blocks = []
    token_count = 0
    doc_count = 0
    for doc_id, tokens in docs:
        counts = {}
        for t in tokens:
            if isinstance(t, str):
                t = t.encode()
            counts[t] = counts.get(t, 0) + 1
        for t in counts:          # first-seen order (dict preserves it)
            if t not in term_ids:
                term_ids[t] = len(terms) + 1
                terms.append(t)
                # idx_terms_add stores token->count (terms.c:262) and
                # dtmap_build_block then increments it again
                # (dtmap.c:228): the introducing doc counts twice.
                totals.append(counts[t])
        pairs = sorted((term_ids[t], c) for t, c in counts.items())
        for tid, c in pairs:
            totals[tid - 1] += c
        blocks.append([doc_id, len(tokens), pairs])
        token_count += len(tokens)
        doc_count += 1
    for rid in removed:
        for blk in blocks:
            if blk[0] == rid and blk[1] != 0:
                for tid, c in blk[2]:
                    totals[tid - 1] -= c
                token_count -= blk[1]
                doc_count -= 1
                blk[0] = 0
                break
        else:
            raise KeyError(rid)
        blocks.append([rid, 0, []])
    return (terms_image(terms, totals),
            dtmap_image([tuple(b) for b in blocks], token_count, doc_count),
            term_ids)


def write_index(basedir, name, docs, removed=(), algo="BM25", filters=(), lang="en"):
    """Create {basedir}/data/{name}/{nxsterms,nxsdtmap,params.db}
    (layout: reference src/core/nxs.c:282-288,421-446)."""
    d = os.path.join(basedir, "data", name)
    os.makedirs(d, exist_ok=True)
    timg, dimg, term_ids = build_images(docs, removed)
    with open(os.path.join(d, "nxsterms"), "wb") as f:
        f.write(timg)
    with open(os.path.join(d, "nxsdtmap"), "wb") as f:
        f.write(dimg)
    flt = ",".join('"%s"' % x for x in filters)
    with open(os.path.join(d, "params.db"), "w") as f:
        f.write('{"algo":"%s","lang":"%s","filters":[%s]}' % (algo, lang, flt))
    return (os.path.join(d, "nxsterms"), os.path.join(d, "nxsdtmap"), term_ids)


def build_images_log(events):
    """Like build_images, from an event log in FILE ORDER: ("add", doc_id,
    tokens) appends a doc block, ("rm", doc_id) zeroes that doc's block id and
    appends a tombstone (dtmap.c:546-658) -- so every prefix of the log is an
    append-only earlier state of the same files (what idx_*_sync consumes)."""
    term_ids = {}
    terms, totals = [], []
    blocks = []
    token_count = 0
    doc_count = 0
    for ev in events:
        if ev[0] == "add":
            _, doc_id, tokens = ev
            counts = {}
            for t in tokens:
                if isinstance(t, str):
                    t = t.encode()
                counts[t] = counts.get(t, 0) + 1
            for t in counts:
                if t not in term_ids:
                    term_ids[t] = len(terms) + 1
                    terms.append(t)
                    totals.append(counts[t])
            pairs = sorted((term_ids[t], c) for t, c in counts.items())
            for tid, c in pairs:
                totals[tid - 1] += c
            blocks.append([doc_id, len(tokens), pairs])
            token_count += len(tokens)
            doc_count += 1
        else:
            rid = ev[1]
            for blk in blocks:
                if blk[0] == rid and blk[1] != 0:
                    for tid, c in blk[2]:
                        totals[tid - 1] -= c
                    token_count -= blk[1]
                    doc_count -= 1
                    blk[0] = 0
                    break
            else:
                raise KeyError(rid)
            blocks.append([rid, 0, []])
    return (terms_image(terms, totals),
            dtmap_image([tuple(b) for b in blocks], token_count, doc_count),
            term_ids)


def publish_in_place(tpath, dpath, timg, dimg):
    """Rewrite both files IN PLACE (same inode, MAP_SHARED readers see it):
    body first, the header -- with data_len -- last, like the reference's
    publish order (terms.c:303-305, dtmap.c:327-337)."""
    for path, img, hdr in ((tpath, timg, 16), (dpath, dimg, 32)):
        with open(path, "r+b") as f:
            f.seek(hdr)
            f.write(img[hdr:])
            f.flush()
            f.seek(0)
            f.write(img[:hdr])
