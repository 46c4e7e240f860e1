"""Compact view of a gfx950 kernel's instruction stream (memory ops, MFMAs, waits, branches) from a hipcc
--save-temps .s file:   python tools/isa_summary.py file.s <mangled-or-substring> [max_lines]"""
import re
import sys

KEY = r"(s_waitcnt|buffer_load|buffer_store|global_load|global_store|global_atomic|ds_read|ds_write|ds_bpermute|s_barrier|v_mfma|s_cbranch|s_branch|s_endpgm|scratch_)"


def main():
    s = open(sys.argv[1]).read()
    pat = sys.argv[2]
    lim = int(sys.argv[3]) if len(sys.argv) > 3 else 120
    for m in re.finditer(r"^(\S*%s\S*):[^\n]*\n(.*?)\n\s*s_endpgm" % re.escape(pat), s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        out, prev, cnt = [], None, 0
        for line in body.split("\n"):
            t = line.strip()
            if t.startswith(".LBB") and t.endswith(":"):
                if cnt:
                    out.append(f"      x{cnt + 1}")
                out.append(t)
                prev, cnt = None, 0
                continue
            if not re.match(KEY, t):
                continue
            t = t.split(";")[0].strip()
            op = t.split()[0]
            if op == prev and not op.startswith("s_"):
                cnt += 1
                continue
            if cnt:
                out[-1] += f"      x{cnt + 1}"
            out.append("  " + t[:76])
            prev, cnt = op, 0
        print(f"== {name}  ({len(out)} summary lines)")
        print("\n".join(out[:lim]))
        break


if __name__ == "__main__":
    main()
