"""`Statistics` accumulator of the text-only trainer (reference: onmt/Trainer.py:26-82); imported by TrainerMultimodal."""
import math
import sys
import time


class Statistics(object):
    def __init__(self, loss=0, n_words=0, n_correct=0):
        self.loss, self.n_words, self.n_correct = loss, n_words, n_correct
        self.n_src_words = 0
        self.start_time = time.time()

    def update(self, stat):
        self.loss += stat.loss
        self.n_words += stat.n_words
        self.n_correct += stat.n_correct

    def accuracy(self):
        return 100 * (self.n_correct / self.n_words)

    def ppl(self):
        return math.exp(min(self.loss / self.n_words, 100))

    def elapsed_time(self):
        return time.time() - self.start_time

    def output(self, epoch, batch, n_batches, start):
        t = self.elapsed_time()
        print("Epoch %2d, %5d/%5d; acc: %6.2f; ppl: %6.2f; %3.0f src tok/s; %3.0f tgt tok/s; %6.0f s elapsed" %
              (epoch, batch, n_batches, self.accuracy(), self.ppl(), self.n_src_words / (t + 1e-5),
               self.n_words / (t + 1e-5), time.time() - start))
        sys.stdout.flush()
