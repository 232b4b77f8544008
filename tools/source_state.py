"""What library a profile describes: sha256 over csrc/ + include/meshenv.h, the library's own hash, and whether the library
is older than its sources.  `python tools/source_state.py` prints it as JSON; `--require-fresh` exits 1 when the shipped .so is
older than any source (tools/profile_round.sh / profile_set.sh refuse to profile such a library).  tools/summarize_profile.py and
summarize_set.py compare the recorded source hash with the working tree and, where .git exists, refuse to summarise when
`git diff --quiet HEAD -- <csrc>` fails: a committed profile always describes a committed kernel."""
import hashlib, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "reinforcementlearning4meshgeneration_amd")
CSRC = os.path.join(PKG, "csrc")


def source_files():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))) + [os.path.join(ROOT, "include", "meshenv.h")]


def source_hash():
    h = hashlib.sha256()
    for f in source_files():
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def state():
    lib = os.environ.get("MESHENV_LIB", os.path.join(PKG, "libmeshenv_hip.so"))
    out = {"source_sha256": source_hash(), "library": os.path.relpath(lib, ROOT)}
    # build.py leaves "<lib>.source" (the hash of the sources it compiled) next to the binary: file times do not survive the
    # copy to the GPU box, the sidecar does
    built_from = None
    if os.path.exists(lib + ".source"):
        built_from = json.load(open(lib + ".source")).get("source_sha256")
    if os.path.exists(lib):
        out["library_sha256"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()
    out["library_built_from"] = built_from
    out["library_older_than_sources"] = built_from != out["source_sha256"]
    return out


def git_state():
    """(head, csrc_clean) or (None, None) without a repository (the GPU box has no .git)."""
    if not os.path.isdir(os.path.join(ROOT, ".git")):
        return None, None
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
    clean = subprocess.run(["git", "-C", ROOT, "diff", "--quiet", "HEAD", "--", os.path.relpath(CSRC, ROOT), "include/meshenv.h"]).returncode == 0
    return head, clean


def check_recorded(recorded, what):
    """Called by the summarisers: the profile's recorded source hash must be the working tree's, and the tree's csrc must be HEAD's."""
    now = source_hash()
    if recorded.get("source_sha256") != now:
        raise SystemExit(f"{what}: profiled sources {str(recorded.get('source_sha256'))[:12]} != working tree {now[:12]} -- re-profile")
    if recorded.get("library_older_than_sources"):
        raise SystemExit(f"{what}: the profiled library was older than its sources -- rebuild and re-profile")
    head, clean = git_state()
    if head is not None and not clean:
        raise SystemExit(f"{what}: csrc/ differs from HEAD -- commit the kernels first, then summarise")
    return dict(recorded, git_head=head)


if __name__ == "__main__":
    st = state()
    print(json.dumps(st))
    if "--require-fresh" in sys.argv and st["library_older_than_sources"]:
        sys.stderr.write("the shipped library is older than csrc/: run python -m reinforcementlearning4meshgeneration_amd.build\n")
        sys.exit(1)
