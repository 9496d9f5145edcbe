* fuzz seed 10266 of tools/fuzz_generated.py: node n8 is only the gate of M2 (tests/test_generated_host.py, test_near_threshold.py)
VDD vdd 0 DC 2.5
R1 n1 n6 8.235e+04
R2 n2 vdd 1500
R3 n3 n9 101.3
R4 n5 n3 384.6
R5 n6 n4 3.109e+04
R6 n7 n5 481.4
R7 n9 n7 676.4
R8 n2 n6 1.554e+04
R9 n5 n1 4.266e+04
R10 n9 0 178.7
R11 n2 n3 2765
R12 vdd n7 440.5
R13 0 n7 434.7
R14 n4 0 4031
R15 n7 n2 1050
R16 vdd n3 204.8
C1 n1 0 2.646e-12
C2 n2 0 1.812e-13
C3 n3 0 1.189e-12
C4 n4 0 1.603e-13
C5 n5 0 4.431e-14
C7 n7 0 1.163e-13
C9 n9 0 4.99e-12
L1 n7 n9 3.575e-09
VIN n3 0 SIN 1.48 0.478 5.075e+08 0
I1 n5 n3 2.09e-05
M1 n2 n7 vdd p 7.08e-6 0.35e-6 1
M2 n1 n8 vdd p 45.7e-6 0.35e-6 1
M3 n9 vdd 0 n 12.6e-6 0.35e-6 2
M4 n6 n3 vdd p 23e-6 0.35e-6 1
.MODEL 1 VT -0.75 MU 5e-2 COX 0.3e-4 LAMBDA 0.05 CJ0 4.0e-14
.MODEL 2 VT 0.83 MU 1.5e-1 COX 0.3e-4 LAMBDA 0.05 CJ0 4.0e-14
.TRAN 1e-10 4e-09
