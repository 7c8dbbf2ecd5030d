"""``get_audio_filenames`` (reference ``data/dataset.py:24-91``): every audio file below a set of directories.

The reference walks the tree with a hand-rolled recursive ``os.scandir``; ``os.walk`` visits directories in the same
(directory-listing, depth-first) order and swallows unreadable entries the same way.  File filters as there: extension in
``exts`` (case-insensitive), not a dot file, and -- only when ``keywords`` are given -- at least one keyword in the lower-cased
name and no archive artefact (``paxheader`` / ``__macosx``) in it.
"""
import os

_ARCHIVE_ARTEFACTS = ("paxheader", "__macosx")


def _wanted(name, exts, keywords):
    lowered = name.lower()
    if name.startswith(".") or os.path.splitext(lowered)[1] not in exts:
        return False
    if keywords is None:
        return True
    return any(k in lowered for k in keywords) and not any(a in lowered for a in _ARCHIVE_ARTEFACTS)


def fast_scandir(directory, ext, keywords=None):
    """-> (all sub-directories, all matching files) below ``directory``."""
    exts = {e.lower() if e.startswith(".") else "." + e.lower() for e in ext}
    keywords = [k.lower() for k in keywords] if keywords else None
    folders, files = [], []
    seen = set()                      # real paths already walked: symlinked sub-folders are followed (as os.scandir + is_dir() does), cycles are not
    for root, dirs, names in os.walk(directory, followlinks=True):
        real = os.path.realpath(root)
        if real in seen:
            dirs[:] = []
            continue
        seen.add(real)
        folders += [os.path.join(root, d) for d in dirs]
        files += [os.path.join(root, n) for n in names if _wanted(n, exts, keywords)]
    return folders, files


def get_audio_filenames(paths, keywords=None, exts=(".wav", ".mp3", ".flac", ".ogg", ".aif", ".opus")):
    if isinstance(paths, str):
        paths = [paths]
    return [f for path in paths for f in fast_scandir(path, list(exts), keywords)[1]]
