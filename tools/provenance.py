"""Where a PMC record comes from.  profiles/traffic.json holds HBM bytes measured with rocprofv3 --pmc on
the builder's GPU box; the driver's bench run cannot collect its own (the profiler has to wrap the program from its start), so
every record carries what it was measured ON - a fingerprint of the kernel and engine sources (thaler-study_amd/csrc: the same
function here and in bench.py, no git needed, so it works on a gpurun box that has no .git), the commit when the caller passed
one (SC_COMMIT), the date, the exact command - and bench.py hands `traffic` out only while the sources still have that
fingerprint."""
import datetime
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "thaler-study_amd", "csrc")


def csrc_fingerprint():
    """sha256 over (relative path, content) of every source under thaler-study_amd/csrc, build products excluded"""
    h = hashlib.sha256()
    for base, dirs, files in sorted(os.walk(CSRC)):
        dirs[:] = sorted(d for d in dirs if d not in ("build", ".pytest_cache", "__pycache__"))
        for f in sorted(files):
            if not f.endswith((".hip", ".hpp", ".inc", ".h")) and f != "Makefile":
                continue
            path = os.path.join(base, f)
            h.update(os.path.relpath(path, CSRC).encode() + b"\0")
            with open(path, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def record(tag, command):
    return {"csrc_sha16": csrc_fingerprint(), "commit": os.environ.get("SC_COMMIT") or None,
            "date": datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%dT%H:%MZ"), "tag": tag, "command": command}
