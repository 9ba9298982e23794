"""CPU probe (NumPy / LAPACK, long double as the yardstick): P~ = A (S Kuu^-1 - I) formed the reference's way (A from two triangular
solves, then A D) and the one-solve way (X = K^ Luu^-T, then X (Luu^-1 D)) at cond(K_uu) = 1e7 (M = 256, lengthscale 4 spacings,
jitter rung 0).  Both are 2.9e-9 of max|P~| from the long-double value -- the common error of D = S Kuu^-1 - I dominates -- so the
one-solve form is not LESS ACCURATE; it is farther from the reference's own ROUNDING (two valid evaluations of it differ by 4e-8 of
max|g_Z| on the GPU at M = 1024 where the two-solve form's differ by 2e-10), and the reference's numbers are the yardstick.
DESIGN 13c.   python tools/ptilde_forms_probe.py"""
import numpy as np, scipy.linalg as sl
np.random.seed(0)
M=256; n=400; h=1.0/(M-1); ell=4*h; var=0.5
Z=np.linspace(0,1,M)[:,None]; X=np.sort(np.random.rand(n,1),0)
def rbf(A,B): 
    d=A[:,None,0]-B[None,:,0]; return var*np.exp(-0.5*d*d/ell**2)
Kuu=rbf(Z,Z); Kuu+= np.eye(M)*var*1e-6   # rung 0 jitter
K=rbf(X,Z)
Lq=np.eye(M)+0.05/np.sqrt(M)*np.tril(np.random.randn(M,M),-1); S=Lq@Lq.T
print("cond", np.linalg.cond(Kuu))
# long double reference
LD=np.longdouble
def chol_ld(A):
    A=A.astype(LD).copy(); n=A.shape[0]; L=np.zeros_like(A)
    for j in range(n):
        d=A[j,j]-np.dot(L[j,:j],L[j,:j]); L[j,j]=np.sqrt(d)
        L[j+1:,j]=(A[j+1:,j]-L[j+1:,:j]@L[j,:j])/L[j,j]
    return L
def fwd_ld(L,B):  # solve L Y = B
    Y=B.astype(LD).copy()
    for j in range(L.shape[0]):
        Y[j]=(Y[j]-L[j,:j]@Y[:j])/L[j,j]
    return Y
def bwd_ld(L,B):  # solve L^T Y = B
    Y=B.astype(LD).copy()
    for j in range(L.shape[0]-1,-1,-1):
        Y[j]=(Y[j]-L[j+1:,j]@Y[j+1:])/L[j,j]
    return Y
Lr=chol_ld(Kuu); 
Ar=bwd_ld(Lr,fwd_ld(Lr,K.T.astype(LD))).T
Kir=bwd_ld(Lr,fwd_ld(Lr,np.eye(M).astype(LD)))
Dr=S.astype(LD)@Kir-np.eye(M)
Pr=Ar@Dr
# double: two-solve
L=np.linalg.cholesky(Kuu)
Xd=sl.solve_triangular(L,K.T,lower=True).T
A=sl.solve_triangular(L,Xd.T,lower=True,trans='T').T
Ki=sl.cho_solve((L,True),np.eye(M)); Ki=np.tril(Ki)+np.tril(Ki,-1).T
D=S@Ki-np.eye(M)
P2=A@D
W2=sl.solve_triangular(L,D,lower=True)
P1=Xd@W2
sc=np.max(np.abs(Pr))
print("max|P|",float(sc),"max|X|",np.abs(Xd).max(),"max|W2|",np.abs(W2).max(),"max|A|",np.abs(A).max(),"max|D|",np.abs(D).max())
print("two-solve err", float(np.max(np.abs(P2-Pr))/sc), " one-solve err", float(np.max(np.abs(P1-Pr))/sc))
# downstream: colsum of P*K weights
w=np.random.randn(n)
g=lambda P:(w[:,None]*P*K).sum(0)
gr=g(Pr.astype(float)); print("colstat two", np.max(np.abs(g(P2)-gr))/np.max(np.abs(gr)), "one", np.max(np.abs(g(P1)-gr))/np.max(np.abs(gr)))
# alternative one-solve: P = (X St - X) Linv-ish?  P = A S Ki - A;  A S Ki = X (Linv S Linv^T) Linv = X Sw Linv
Sw=sl.solve_triangular(L, sl.solve_triangular(L,S,lower=True).T, lower=True).T   # Linv S Linv^T
Y=Xd@(Sw-np.eye(M))       # n x M  : X (Sw - I)
P3=sl.solve_triangular(L,Y.T,lower=True,trans='T').T   # Y Linv  (backward row-solve)  -> this needs a second n x M solve: only for comparison
print("alt err", float(np.max(np.abs(P3-Pr))/sc))
