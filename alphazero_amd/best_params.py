"""Bookkeeping of the "best" network around the arena: which parameter file the
self-play workers use, and its replacement when the latest network wins the
match (src/eval.py:176-182 with src/tools.py:245-313).

The pointer file keeps the reference's format -- one JSON object {"path": <file>}
(the reference's `.param_best`, general_config.py) -- so either code base can
read what the other wrote.
"""
import glob
import json
import os


class BestParams:
    def __init__(self, pointer_file, param_dir, pattern="*.pyt", dry_run=None):
        """pointer_file: the JSON pointer (PARAM_BEST); param_dir: where parameter files
        live (PARAM_PATH); pattern: glob of parameter files inside param_dir (the
        reference uses "<MODEL_CLASS><SHAPE>_*.pyt", src/tools.py:249-250); dry_run:
        leave the pointer alone (default: the DRY_RUN environment switch, src/tools.py:236-243)."""
        self.pointer_file, self.param_dir, self.pattern = pointer_file, param_dir, pattern
        self._dry_run = dry_run

    def dry_run(self):
        return ("DRY_RUN" in os.environ) if self._dry_run is None else bool(self._dry_run)

    def init_file(self):
        """src/tools.py:310-313: make sure the pointer file exists."""
        open(self.pointer_file, "a").close()

    def get_latest(self):
        """src/tools.py:245-257: the newest parameter file by name order, None without any."""
        files = sorted(glob.glob(os.path.join(self.param_dir, self.pattern)))
        if not files:
            return None
        last = files[-1]
        return last if os.path.isabs(last) else "{}/{}".format(self.param_dir, last)

    def get_best(self):
        """src/tools.py:271-292: the recorded best; an empty pointer adopts the latest
        file (and records it); a recorded file that has moved is looked up by name
        inside param_dir and that path is returned whether or not it exists."""
        with open(self.pointer_file, "r+") as fp:
            try:
                content = json.load(fp)
            except ValueError:
                path = self.get_latest()
                if path is None:
                    return None
                self.set_best(path)
                return path
        if not os.path.isfile(content["path"]):
            path = os.path.join(self.param_dir, os.path.basename(content["path"]))
            if not os.path.isfile(path):
                print("Could not found parameters")
                print("Running with default parameters")
            return path
        return content["path"]

    def set_best(self, path):
        """src/tools.py:294-308."""
        if self.dry_run():
            print("Dry run: setting {} as best".format(path))
            return
        if not os.path.isabs(path):
            path = "{}/{}".format(self.param_dir, path)
        if not os.path.isfile(path):
            raise RuntimeError("File {} does not exst".format(path))
        with open(self.pointer_file, "w") as fp:
            json.dump({"path": os.path.normpath(path)}, fp)


def conclude_match(book, latest_path, wins_best, wins_latest, threshold=0.6, on_promote=None):
    """End of eval_models (src/eval.py:176-182): the latest network becomes the best when
    wins_latest / (wins_latest + wins_best) >= threshold (EVAL_TRESHOLD,
    general_config.py:24); draws do not count, a match without a decided game divides
    by zero as in the reference.  on_promote: what follows a promotion there
    (tools.init_generation: a new data generation).  Returns True on promotion."""
    if wins_latest / (wins_latest + wins_best) >= threshold:
        book.set_best(latest_path)
        if on_promote is not None:
            on_promote()
        return True
    return False
