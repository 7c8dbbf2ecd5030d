"""``get_audio_filenames`` (reference ``data/dataset.py:24-91``): recursive directory scan."""
import os


def fast_scandir(directory, ext, keywords=None):
    subfolders, files = [], []
    ext = ["." + x if x[0] != "." else x for x in ext]
    bad_prefixes = ["."]
    keywords = [k.lower() for k in keywords] if keywords else None
    banned = ("paxheader", "__macosx")     # only consulted when keywords are given (reference quirk)
    try:
        for f in os.scandir(directory):
            try:
                if f.is_dir():
                    subfolders.append(f.path)
                elif f.is_file():
                    file_ext = os.path.splitext(f.name)[1].lower()
                    is_hidden = any(os.path.basename(f.path).startswith(p) for p in bad_prefixes)
                    has_ext = file_ext in ext
                    name_lower = f.name.lower()
                    has_keyword = any(k in name_lower for k in keywords) if keywords else True
                    has_banned = any(w in name_lower for w in banned) if keywords else False
                    if has_ext and has_keyword and not is_hidden and not has_banned:
                        files.append(f.path)
            except Exception:
                pass
    except Exception:
        pass
    for d in list(subfolders):
        sf, f = fast_scandir(d, ext, keywords)
        subfolders.extend(sf)
        files.extend(f)
    return subfolders, files


def get_audio_filenames(paths, keywords=None, exts=(".wav", ".mp3", ".flac", ".ogg", ".aif", ".opus")):
    filenames = []
    if isinstance(paths, str):
        paths = [paths]
    for p in paths:
        _, files = fast_scandir(p, list(exts), keywords)
        filenames.extend(files)
    return filenames
