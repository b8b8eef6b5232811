#!/usr/bin/env python3
"""Round 6: where a TF step's wall time sits at its END -- per step, from a rocprofv3 --kernel-trace CSV: the last kernel of the backward on the main
queue, the last weight-gradient / reduce kernel on the side queue, the optimiser launches; and the same at the decoder -> encoder hand-over.
python3 scripts/r6/tail_analysis.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
lo_ms, hi_ms = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.0, 1e9)      # keep the steps whose loss-to-loss wall time is in this range
name = lambda r: r["Kernel_Name"]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
ce = [i for i, r in enumerate(rows) if "softmax_ce" in name(r)]
print("steps:", len(ce), "queue column:", qkey, "queues:", collections.Counter(r[qkey] for r in rows).most_common(6))
mainq = rows[ce[len(ce) // 2]][qkey]
out = []
for a, b in zip(ce[3:-1], ce[4:]):
    if not (lo_ms <= (rows[b]["s"] - rows[a]["s"]) / 1e6 <= hi_ms):
        continue
    step = rows[a:b]                                  # softmax_ce of step i .. softmax_ce of step i+1: backward i, optimiser i, forward i+1
    adam = [r for r in step if "adamw" in name(r)]
    if not adam:
        continue
    last_adam = max(adam, key=lambda r: r["e"])
    fin = last_adam["e"]
    before = [r for r in step if r["e"] <= last_adam["s"] + 1 or r["s"] < last_adam["s"]]
    main = [r for r in before if r[qkey] == mainq and "adamw" not in name(r)]
    side = [r for r in before if r[qkey] != mainq]
    lm, ls = max(main, key=lambda r: r["e"]), max(side, key=lambda r: r["e"])
    side_tail = [r for r in side if r["e"] > lm["e"]]
    out.append(dict(main_end=lm["e"], side_end=ls["e"], adam_s=last_adam["s"], adam_e=fin, lm=name(lm)[:40], ls=name(ls)[:40],
                    tail=[(name(r)[:34], (r["s"] - lm["e"]) / 1e3, (r["e"] - lm["e"]) / 1e3) for r in side_tail], n_adam=len(adam),
                    adams=[((r["s"] - lm["e"]) / 1e3, (r["e"] - lm["e"]) / 1e3, r[qkey] == mainq) for r in adam]))
for o in out[:4]:
    print(f"last main kernel {o['lm']} ends at 0; last side kernel {o['ls']} ends at {(o['side_end'] - o['main_end']) / 1e3:8.1f} us; "
          f"last AdamW runs {(o['adam_s'] - o['main_end']) / 1e3:8.1f} .. {(o['adam_e'] - o['main_end']) / 1e3:8.1f} us")
    print("   AdamW launches (start, end, on main queue):", [(round(a, 1), round(b, 1), c) for a, b, c in o["adams"]])
    for t in o["tail"][-12:]:
        print(f"   side after main's end: {t[0]:36s} {t[1]:8.1f} .. {t[2]:8.1f} us")
import statistics
print("median exposed tail (last main kernel end -> last AdamW end), us:", statistics.median((o["adam_e"] - o["main_end"]) / 1e3 for o in out))
print("median (last side kernel end - last main kernel end), us:", statistics.median((o["side_end"] - o["main_end"]) / 1e3 for o in out))
