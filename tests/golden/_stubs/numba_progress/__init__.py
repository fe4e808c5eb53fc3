class ProgressBar:
    def __init__(self, total=None, disable=True, **kwargs):
        self.total = total

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def update(self, n=1):
        return None
