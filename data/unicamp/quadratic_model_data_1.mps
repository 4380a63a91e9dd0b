************************************************************************
*
*  The data in this file represents the quadratic matrix for
*  the following problem:
*
*  Minimize Z = x1 + 2x5 - x8 +
*               1/2(x1**2 + x2**2 + x3**2 + x4**2 +
*                   x5**2 + x6**2 + x7**2 + x8**2)
*
*  where the linear part of the problem is in "Sample Linear Programming
*  Model Data 1".
*
************************************************************************
NAME          EXAMPLE
QSECTION
    COL01     COL01       1.0000D+00
    COL02     COL02       1.0000D+00
    COL03     COL03       1.0000D+00
    COL04     COL04       1.0000D+00
    COL05     COL05       1.0000D+00
    COL06     COL06       1.0000D+00
    COL07     COL07       1.0000D+00
    COL08     COL08       1.0000D+00
ENDATA
