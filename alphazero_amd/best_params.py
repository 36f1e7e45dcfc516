"""Which parameter file is "the best network", kept across arena matches.

Self-play workers ask for the best network, the arena replaces it when the
latest network wins a match (the role of src/tools.py:245-313 around
src/eval.py:176-182).  State is one small JSON document, {"path": <file>},
the layout of the reference's `.param_best`, so either code base can read what
the other wrote.

Behaviour, stated once:
  * no pointer yet        -> the newest parameter file is adopted and recorded;
  * pointer to a file that is gone (the directory moved) -> the file of the same
    name inside the parameter directory, existing or not (the caller then
    starts from untrained weights, as the reference does);
  * a dry run never writes.
"""
import json
import os
from pathlib import Path


class BestParams:
    def __init__(self, pointer_file, param_dir, pattern="*.pyt", dry_run=None):
        """pointer_file: the JSON pointer; param_dir: directory of parameter files;
        pattern: glob for them inside param_dir ("<model><shape>_*.pyt" in the
        reference's naming); dry_run: True/False, or None = follow the DRY_RUN
        environment switch the reference's scripts use."""
        self.pointer_file = str(pointer_file)
        self.param_dir = str(param_dir)
        self.pattern = pattern
        self._dry_run = dry_run

    # -- small pieces ------------------------------------------------------
    def dry_run(self):
        if self._dry_run is not None:
            return bool(self._dry_run)
        return "DRY_RUN" in os.environ

    def init_file(self):
        """Create an empty pointer if there is none."""
        Path(self.pointer_file).touch()

    def _in_dir(self, name):
        """A bare or relative name means a file of the parameter directory."""
        return name if os.path.isabs(name) else f"{self.param_dir}/{name}"

    def _recorded(self):
        """The path the pointer holds, None for an empty or unreadable pointer."""
        try:
            doc = json.loads(Path(self.pointer_file).read_text())
        except ValueError:
            return None
        return doc.get("path") if isinstance(doc, dict) else None

    # -- queries -----------------------------------------------------------
    def get_latest(self):
        """Newest parameter file = last in name order (names carry a zero-padded
        counter); None when nothing has been trained yet."""
        names = sorted(p.name for p in Path(self.param_dir).glob(self.pattern))
        return self._in_dir(names[-1]) if names else None

    def get_best(self):
        recorded = self._recorded()
        if recorded is None:
            newest = self.get_latest()
            if newest is not None:
                self.set_best(newest)
            return newest
        if os.path.isfile(recorded):
            return recorded
        moved = os.path.join(self.param_dir, os.path.basename(recorded))
        if not os.path.isfile(moved):
            print(f"best parameters {recorded} not found, {moved} neither: starting from untrained weights")
        return moved

    # -- update ------------------------------------------------------------
    def set_best(self, path):
        if self.dry_run():
            print(f"dry run: {path} would become the best parameters")
            return
        target = self._in_dir(path)
        if not os.path.isfile(target):
            raise RuntimeError(f"parameter file {target} does not exist")
        Path(self.pointer_file).write_text(json.dumps({"path": os.path.normpath(target)}))


def conclude_match(book, latest_path, wins_best, wins_latest, threshold=0.6, on_promote=None):
    """Verdict of an arena match (src/eval.py:176-182): the latest network is promoted
    when its share of the DECIDED games reaches `threshold` (0.6 in
    general_config.py:24).  Draws do not count; a match without a decided game
    divides by zero, as in the reference.  on_promote runs after a promotion
    (the reference opens a new data generation there).  Returns True on promotion."""
    share = wins_latest / (wins_latest + wins_best)
    if share < threshold:
        return False
    book.set_best(latest_path)
    if on_promote is not None:
        on_promote()
    return True
