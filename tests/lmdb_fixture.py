"""Test-only writer of an LMDB data file (one committed transaction, main database only) following liblmdb's on-disk
layout: meta pages 0 / 1, leaf pages (nodes sorted by key, big values on overflow pages), one level of branch pages when
the leaves do not fit one page.  It exists because neither liblmdb nor py-lmdb is available in the build image; the
product only READS LMDB files (rick_amd.data.LmdbReader)."""
import os
import struct

PSIZE = 4096
HDR = 16
P_BRANCH, P_LEAF, P_OVERFLOW, P_META = 1, 2, 4, 8
F_BIGDATA = 1


def _page(pgno, flags, body_nodes):
    """body_nodes: list of node byte strings; nodes are packed from the page end downwards (even offsets)."""
    page = bytearray(PSIZE)
    upper = PSIZE
    ptrs = []
    for nd in body_nodes:
        size = (len(nd) + 1) & ~1
        upper -= size
        page[upper:upper + len(nd)] = nd
        ptrs.append(upper)
    lower = HDR + 2 * len(ptrs)
    assert lower <= upper, 'page overflow'
    struct.pack_into('<QHHHH', page, 0, pgno, 0, flags, lower, upper)
    for i, p in enumerate(ptrs):
        struct.pack_into('<H', page, HDR + 2 * i, p)
    return bytes(page)


def write_lmdb(path, items, max_inline=1000):
    """items: dict bytes -> bytes.  Writes <path>/data.mdb."""
    os.makedirs(path, exist_ok=True)
    keys = sorted(items)
    pages = {}
    next_pg = 2
    leaf_nodes = []
    overflow_pages = 0
    for k in keys:
        v = items[k]
        if len(v) > max_inline:
            n_ovf = (HDR + len(v) + PSIZE - 1) // PSIZE
            first = next_pg
            next_pg += n_ovf
            overflow_pages += n_ovf
            blob = bytearray(n_ovf * PSIZE)
            struct.pack_into('<QHHI', blob, 0, first, 0, P_OVERFLOW, n_ovf)
            blob[HDR:HDR + len(v)] = v
            for j in range(n_ovf):
                pages[first + j] = bytes(blob[j * PSIZE:(j + 1) * PSIZE])
            node = struct.pack('<HHHH', len(v) & 0xffff, len(v) >> 16, F_BIGDATA, len(k)) + k + struct.pack('<Q', first)
        else:
            node = struct.pack('<HHHH', len(v) & 0xffff, len(v) >> 16, 0, len(k)) + k + v
        leaf_nodes.append((k, node))
    # pack leaves greedily
    leaves, cur, used = [], [], HDR
    for k, node in leaf_nodes:
        need = ((len(node) + 1) & ~1) + 2
        if used + need > PSIZE and cur:
            leaves.append(cur)
            cur, used = [], HDR
        cur.append((k, node))
        used += need
    if cur:
        leaves.append(cur)
    leaf_pgnos = []
    for lf in leaves:
        pages[next_pg] = _page(next_pg, P_LEAF, [nd for _, nd in lf])
        leaf_pgnos.append((lf[0][0], next_pg))
        next_pg += 1
    depth, branch_pages = 1, 0
    root = leaf_pgnos[0][1] if leaf_pgnos else (1 << 64) - 1
    if len(leaf_pgnos) > 1:
        nodes = []
        for i, (k, pg) in enumerate(leaf_pgnos):
            key = b'' if i == 0 else k
            nodes.append(struct.pack('<HHHH', pg & 0xffff, (pg >> 16) & 0xffff, (pg >> 32) & 0xffff, len(key)) + key)
        pages[next_pg] = _page(next_pg, P_BRANCH, nodes)
        root, depth, branch_pages = next_pg, 2, 1
        next_pg += 1
    last_pg = next_pg - 1

    def meta(pgno, txnid):
        page = bytearray(PSIZE)
        struct.pack_into('<QHHHH', page, 0, pgno, 0, P_META, 0, 0)
        off = HDR
        struct.pack_into('<IIQQ', page, off, 0xBEEFC0DE, 1, 0, 1 << 30)
        off += 24
        struct.pack_into('<IHHQQQQQ', page, off, PSIZE, 0, 0, 0, 0, 0, 0, (1 << 64) - 1)          # free DB (md_pad = page size)
        off += 48
        struct.pack_into('<IHHQQQQQ', page, off, 0, 0, depth, branch_pages, len(leaves), overflow_pages, len(keys), root)
        off += 48
        struct.pack_into('<QQ', page, off, last_pg, txnid)
        return bytes(page)
    with open(os.path.join(path, 'data.mdb'), 'wb') as f:
        f.write(meta(0, 0))          # older meta (empty environment would have root = invalid; content is never read)
        f.write(meta(1, 1))          # the committed transaction
        for pg in range(2, next_pg):
            f.write(pages[pg])
