import sys, os, warnings
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pycmf_amd import CMF
g = np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/g4_fit_level.npz"))
warnings.simplefilter("ignore")
for solver in ("mu", "newton"):
    m = CMF(n_components=5, solver=solver, x_init="custom", y_init="custom", random_state=0, max_iter=1000)
    U, V, Z = m.fit_transform(g["fc_X"], g["fc_Y"], U=g["fc_U0"].copy(), V=g["fc_V0"].copy(), Z=g["fc_Z0"].copy())
    print(solver, "n_iter gpu %d ref %d | err gpu %.8f ref %.8f rel %.2e | max|U-Uref| %.2e" % (m.n_iter_, int(g["fc_%s_n_iter" % solver]), m.reconstruction_err_, float(g["fc_%s_err" % solver]), abs(m.reconstruction_err_ - float(g["fc_%s_err" % solver])) / float(g["fc_%s_err" % solver]), np.abs(U - g["fc_%s_U" % solver]).max()))
m = CMF(n_components=5, solver="newton", y_link="logit", random_state=42, max_iter=200, U_non_negative=False, V_non_negative=False, Z_non_negative=False)
m.fit(g["lg_X"], g["lg_Y"])
print("logit n_iter gpu %d ref %d | err gpu %.8f ref %.8f" % (m.n_iter_, int(g["lg_n_iter"]), m.reconstruction_err_, float(g["lg_err"])))
g1 = np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/g1_readme.npz"))
m = CMF(n_components=4, random_state=0); m.fit(g1["X"], g1["Y"])
print("readme n_iter gpu %d ref %d | err gpu %.8f ref %.8f" % (m.n_iter_, int(g1["n_iter"]), m.reconstruction_err_, float(g1["err"])))
