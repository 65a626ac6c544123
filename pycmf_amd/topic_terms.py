"""Topic-term printing helpers: presentation only (pycmf/analysis.py:3-16).

Output format follows the reference: topics are numbered from 1, the ten
highest-weight terms are listed in ascending weight order (``argsort()[-10:]``;
``topn_words`` is accepted but, like upstream, not applied), and the
"with importances" variant prints the topic's label weights with 3 decimals.
"""
import numpy as np


def _top_terms(column, idx_to_word):
    order = np.argsort(column)[-10:]
    return ",".join(str(w) for w in np.asarray(idx_to_word)[order])


def print_topic_terms_from_matrix(term_topic, idx_to_word, topn_words=10, n_topics=100):
    for t in range(min(term_topic.shape[1], n_topics)):
        print("Topic {}: {}".format(t + 1, _top_terms(term_topic[:, t], idx_to_word)))


def print_topic_terms_with_importances(term_topic, label_topic, idx_to_word, topn_words=10, n_topics=100):
    n = min(term_topic.shape[1], label_topic.shape[1], n_topics)
    for t in range(n):
        weights = ",".join("{:.3f}".format(w) for w in label_topic[:, t])
        print("Topic {} [{}]: {}".format(t + 1, weights, _top_terms(term_topic[:, t], idx_to_word)))
