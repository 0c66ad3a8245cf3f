#!/usr/bin/env python3
"""Regenerate the fixtures of tests/test_tools.py: a small rnamotif output (made by the
test-only oracle CLI, which is byte-identical to the reference on these descriptors)
run through the REFERENCE's own rm2ct, rmfmt and rmprune (oracle/_ref/, built from
/root/reference/src as it lies).  Needs /root/reference; run from the repo root."""
import gzip
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
HERE = os.path.join(ROOT, "tests", "golden", "tools")
REF = os.path.join(ROOT, "oracle", "_ref")
ENV = dict(os.environ, LC_ALL="C", EFNDATA=os.path.join(ROOT, "rnamotif_amd", "efndata"))


def run(cmd, data=None, cwd=None):
    return subprocess.run(cmd, input=data, cwd=cwd, env=ENV, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                          check=True).stdout


def main():
    work = sys.argv[1] if len(sys.argv) > 1 else "/tmp/tool_goldens"
    os.makedirs(work, exist_ok=True)
    db = os.path.join(work, "gbrna.111.0.fastn")
    with gzip.open(os.path.join(ROOT, "tests", "golden", "test", "gbrna.111.0.fastn.gz"), "rb") as f:
        open(db, "wb").write(f.read())
    for name in ("trna", "pk1", "score.2"):
        src = os.path.join(ROOT, "tests", "golden", "test", name + ".descr")
        out = run([os.path.join(ROOT, "oracle", "rnamotif_oracle"), "-descr", name + ".descr", db],
                  cwd=os.path.dirname(src))
        # header + the hits of the first 12 database entries that have any
        lines = out.split(b"\n")
        keep, entries, last = [], 0, None
        for i, l in enumerate(lines):
            if l.startswith(b">"):
                sid = l.split()[0]
                if sid != last:
                    entries += 1
                    last = sid
                if entries > 12:
                    break
            keep.append(l)
        sample = b"\n".join(keep) + b"\n"
        open(os.path.join(HERE, name + ".rm.out"), "wb").write(sample)
        for tool, opts in (("rm2ct", []), ("rm2ct", ["-t", "rnaviz"]), ("rmfmt", []), ("rmfmt", ["-l"]),
                           ("rmfmt", ["-la"]), ("rmfmt", ["-a"]), ("rmfmt", ["-a", "-l"]), ("rmprune", [])):
            got = run([os.path.join(REF, tool)] + opts, data=sample)
            tag = "".join(o.strip("-") for o in opts)
            open(os.path.join(HERE, "%s.%s%s.ref" % (name, tool, "." + tag if tag else "")), "wb").write(got)


if __name__ == "__main__":
    main()
