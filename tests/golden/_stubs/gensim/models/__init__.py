class Word2Vec:  # placeholder: embedding training is out of scope for the goldens
    def __init__(self, *args, **kwargs):
        raise RuntimeError("gensim stub: Word2Vec is not available in the golden harness")
