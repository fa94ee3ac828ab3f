"""Turns the file tests/conftest.py writes under SLAM_TEST_MARGINS=<path> into a markdown table: per test, the worst share of an
allowed cosine deviation that a measurement used.  Rule (VERDICT r4 next #5c): no bound within 2x of a measured value, i.e. share <= 0.5.

    SLAM_TEST_MARGINS=gpurun_out/margins.tsv python -m pytest tests -m gpu -q ; python tools/margins_report.py gpurun_out/margins.tsv
"""
import collections
import sys

rows = [l.rstrip("\n").split("\t") for l in open(sys.argv[1])][1:]
worst = collections.OrderedDict()
for t, what, dev, allowed in rows:
    dev, allowed = float(dev), float(allowed)
    share = dev / allowed if allowed > 0 else float("inf")
    if t not in worst or share > worst[t][0]:
        worst[t] = (share, what, dev, allowed, 0)
    worst[t] = worst[t][:4] + (worst[t][4] + 1,)
print("| test | floors checked | worst share of the allowed 1 - cos used | measured 1 - cos | allowed | which |")
print("|---|---|---|---|---|---|")
over = 0
for t, (share, what, dev, allowed, n) in sorted(worst.items(), key=lambda kv: -kv[1][0]):
    flag = " **> 0.5**" if share > 0.5 else ""
    over += share > 0.5
    print(f"| `{t.replace('tests/', '')}` | {n} | {share:.2f}{flag} | {dev:.2e} | {allowed:.2e} | {what.split(':')[0][:70]} |")
print()
print(f"{len(rows)} cosine floors in {len(worst)} tests; {over} test(s) use more than half of an allowed deviation")
