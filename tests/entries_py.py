"""Test infrastructure: the 48-byte insert-candidate entries of include/tcmi.h (TCMI_INS_ENTRY_BYTES; csrc/tcmi_internal.h
tcmi_dev_entry, written on the device by pack_device.hip's ins_entries_kernel), restated in plain Python from flat read arrays — so
that the exchange of configs[4] (distributed.consensus_split_bamfile: pieces per rank, concatenated per column in rank order, voted on
by tcmi_modal_from_entries) can be tested on a box without a GPU, against the oracle's own vote (oracle/tc_oracle.py region_tokens,
which follows pysam's default region pile-up behind Events.py:47-82)."""
import numpy as np

ENT = np.dtype([("key", "<u8"), ("name_hash", "<u8"), ("j", "<u4"), ("pos", "<i4"), ("end", "<i4"), ("mpos", "<i4"), ("isize", "<i4"),
                ("l_qseq", "<i4"), ("flag", "<u2"), ("qual", "u1"), ("bits", "u1"), ("qref", "<i4")])
assert ENT.itemsize == 48
NT = "=ACMGRSVTWYHKDBN"
REF = (0, 2, 3, 7, 8)
MATCH = (0, 7, 8)


def entries_for(reads, positions, flag_filter=0x4 | 0x100 | 0x200 | 0x400, ignore_orphans=True, j0=0):
    """-> (entry bytes, ent_off list [n_pos + 1], long-insertion text bytes): per 1-based position the entries of the reads that
    cover it, in the order of the arrays (file order)."""
    n = int(reads["n_reads"])
    co, so = np.asarray(reads["cigar_off"], np.int64), np.asarray(reads["seq_off"], np.int64)
    lq_all = np.asarray(reads["l_qseq"], np.int64)
    qo = np.concatenate(([0], np.cumsum(lq_all)))
    qual = reads.get("qual")
    out, off, text = [], [0], bytearray()
    spans = []
    for i in range(n):
        cg = np.asarray(reads["cigar"][co[i]:co[i + 1]], np.int64)
        spans.append(int(sum(int(c) >> 4 for c in cg if (int(c) & 15) in REF)))
    for p1 in positions:
        col = int(p1) - 1
        rows = []
        for i in range(n):
            fl = int(reads["flag"][i])
            pos = int(reads["pos"][i])
            tid = int(reads["tid"][i]) if reads.get("tid") is not None else 0
            if (fl & 4) or tid != 0 or pos < 0 or spans[i] == 0 or not (pos <= col < pos + spans[i]):
                continue
            if fl & flag_filter or (ignore_orphans and (fl & 1) and not (fl & 2)):
                continue
            cg = [int(c) for c in reads["cigar"][co[i]:co[i + 1]]]
            seq = reads["seq"][so[i]:so[i + 1]]
            lq = int(lq_all[i])
            nib = lambda q: 15 if q >= lq else (int(seq[q >> 1]) & 15 if q & 1 else int(seq[q >> 1]) >> 4)
            x, y = pos, 0
            for c, w in enumerate(cg):
                op, ln = w & 15, w >> 4
                if op in REF:
                    if col < x + ln:
                        rev = bool(fl & 0x10)
                        qpos = y + (col - x) if op in MATCH else y
                        e = np.zeros(1, ENT)[0]
                        e["qual"] = (int(qual[qo[i] + qpos]) if qual is not None else 255) if qpos < lq else 0
                        nb = nib(qpos)
                        bits = nb | (0x10 if op in MATCH else 0)
                        first = NT[nb] if op in MATCH else (("<" if rev else ">") if op == 3 else "*")
                        if first == "=":
                            first = "," if rev else "."
                        indel = 0
                        if col == x + ln - 1 and c + 1 < len(cg):
                            op2 = cg[c + 1] & 15
                            if op2 == 2 and op != 2:
                                indel = -(cg[c + 1] >> 4)
                                for t in cg[c + 2:]:
                                    if t & 15 != 2:
                                        break
                                    indel -= t >> 4
                            elif op2 == 1:
                                indel = cg[c + 1] >> 4
                                for t in cg[c + 2:]:
                                    if t & 15 == 1:
                                        indel += t >> 4
                                    elif t & 15 != 6:
                                        break
                            elif op2 == 6 and c + 2 < len(cg):
                                for t in cg[c + 2:]:
                                    if t & 15 == 1:
                                        indel += t >> 4
                                    elif (t & 15) in REF:
                                        break
                        key = (1 << 63) | ord(first)
                        if indel > 12:
                            bits |= 0x40
                            slot = len(text)
                            text.extend(nib(qpos + t) for t in range(1, indel + 1))
                            key |= (slot << 8) | (indel << 40)
                        elif indel > 0:
                            key |= (1 << 8) | (indel << 10)
                            any_eq = False
                            for t in range(1, indel + 1):
                                nbt = nib(qpos + t)
                                any_eq |= nbt == 0
                                key |= nbt << (15 + 4 * (t - 1))
                            if any_eq and rev:
                                key |= 1 << 14
                        elif indel < 0:
                            key |= (2 << 8) | ((-indel) << 10)
                        e["key"] = key
                        mt = int(reads["next_tid"][i]) if reads.get("next_tid") is not None else -1
                        e["mpos"] = int(reads["next_pos"][i]) if reads.get("next_pos") is not None else -1
                        e["isize"] = int(reads["tlen"][i]) if reads.get("tlen") is not None else 0
                        if mt >= 0 and mt != tid:
                            bits |= 0x20
                        h = 1469598103934665603
                        if reads.get("names") is not None and reads.get("name_off") is not None:
                            no = reads["name_off"]
                            name = bytes(np.asarray(reads["names"][int(no[i]):int(no[i + 1])], np.uint8))
                        else:
                            name = b"r%d" % (j0 + i)
                        for ch in name:
                            h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
                        e["name_hash"] = h or 1
                        e["j"], e["pos"], e["end"], e["l_qseq"], e["flag"], e["bits"] = j0 + i, pos, pos + spans[i], lq, fl, bits
                        qref = -1
                        if op not in MATCH and qpos < lq:
                            xr = x + ln
                            for t in cg[c + 1:]:
                                o, l2 = t & 15, t >> 4
                                if o in MATCH:
                                    if l2 > 0:
                                        qref = xr
                                    break
                                if o in (1, 4) and l2 > 0:
                                    break
                                if o in REF:
                                    xr += l2
                        e["qref"] = qref
                        rows.append(e)
                        break
                    x += ln
                if op in (0, 1, 4, 7, 8):
                    y += ln
        out.extend(rows)
        off.append(off[-1] + len(rows))
    arr = np.array(out, ENT) if out else np.zeros(0, ENT)
    return arr.tobytes(), off, bytes(text)
