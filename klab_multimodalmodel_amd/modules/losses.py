"""Per-epoch loss bookkeeping (ref/modules/losses.py:4-31): running sums of `loss.item()`, epoch means, a PNG."""
import os


class LossCounter:
    def __init__(self, train_len, val_len):
        self.len = {'train': train_len, 'val': val_len}
        self.running = {'train': 0.0, 'val': 0.0}
        self.losses = {'train': [], 'val': []}

    def add_loss(self, phase, loss):
        self.running[phase] += loss

    def count_and_get_loss(self):
        out = []
        for phase in ('train', 'val'):
            mean = self.running[phase] / self.len[phase]
            self.losses[phase].append(mean)
            self.running[phase] = 0.0
            out.append(mean)
        return tuple(out)

    def plot_loss(self, result_dir):
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        plt.figure()
        for phase in ('train', 'val'):
            plt.plot(self.losses[phase], label=phase.capitalize())
        plt.xlabel('Epoch')
        plt.ylabel('Loss')
        plt.legend()
        plt.savefig(os.path.join(result_dir, "loss.png"))
        plt.close()
