"""Optional visdom curves (jTransUP/utils/visuliazer.py).  visdom is not part of the accelerated path; when the
package or the server is missing, plotting degrades to a no-op instead of failing the run."""


class Visualizer(object):
    def __init__(self, env='default', port=8097, **kwargs):
        self.index, self.log_text, self.vis = {}, '', None
        try:
            import visdom
            self.vis = visdom.Visdom(env=env, port=port, **kwargs)
        except Exception:      # noqa: BLE001 -- optional UI
            self.vis = None

    def plot_many_stack(self, points, win_name='', options=None):
        if self.vis is None:
            return
        import numpy as np
        names, vals = list(points.keys()), list(points.values())
        y = np.array(vals).reshape(-1, len(vals)) if len(vals) > 1 else np.array(vals)
        x = self.index.get(win_name, 0)
        self.vis.line(Y=y, X=np.ones(y.shape) * x, win=win_name, opts=dict(legend=names, title=win_name),
                      update=None if x == 0 else 'append')
        self.index[win_name] = x + 1

    def log(self, info, win_name='log_text'):
        if self.vis is None:
            return
        self.log_text += '{}<br>'.format(info)
        self.vis.text(self.log_text, win_name)
