"""Synthetic (Question, Abstract, Answer) rows as SURVEY.md section 8(d) specifies: fixed word list,
random.Random(1234); Question 4-12 words, Abstract 20-110 words, Answer 1-8 words.

    python tools/make_synthetic_csv.py out.csv --rows 200000      # cfg5 scale (the reference README's 200 k)
"""
from __future__ import annotations

import argparse
import csv
import random

SYLL = ["ka", "lo", "mi", "ren", "tu", "vas", "ze", "pha", "dro", "quin", "sol", "ber", "ny", "ox", "ume", "tal"]


def word_list(n: int = 4000, seed: int = 7):
    r = random.Random(seed)
    words = set()
    while len(words) < n:
        words.add("".join(r.choice(SYLL) for _ in range(r.randint(1, 4))))
    return sorted(words)


def write_csv(path: str, rows: int, seed: int = 1234) -> None:
    words = word_list()
    r = random.Random(seed)

    def text(lo, hi):
        return " ".join(r.choice(words) for _ in range(r.randint(lo, hi)))

    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Question", "Abstract", "Answer"])
        for _ in range(rows):
            w.writerow([text(4, 12), text(20, 110), text(1, 8)])


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--rows", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=1234)
    a = ap.parse_args()
    write_csv(a.path, a.rows, a.seed)
    print(f"wrote {a.rows} rows to {a.path}")


if __name__ == "__main__":
    main()
